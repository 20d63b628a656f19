// Micro-test for the "3-way bf16 split" of the hidden 32 x 64 transform (round-3 review, item 6 (i)).
// H = h1 + h2 + h3 and W = w1 + w2 + w3 exactly (three bf16 each cover float32's 24 mantissa bits), every bf16 x bf16
// product is exact in float32, so H.W = sum of 9 (or the 6 largest) bf16 MFMA terms with float32 accumulation - at 16 x the
// fp32 MFMA rate per instruction.  Two questions decide whether it can replace v_mfma_f32_16x16x4_f32 in a library whose
// CPU twin mirrors every kernel bit for bit:
//   (1) WHAT does v_mfma_f32_16x16x16_bf16 compute - which of: a sequential float32 fma chain over k; the exact sum of the 16
//       products (+ C) rounded once; four exact 4-term group sums added in order; a pairwise tree ...  (the twin must do the same)
//   (2) what do 24 / 36 bf16 MFMAs (6 / 9 terms x 4 column tiles, k = 16 per instruction, two per 32-deep tile) cost next to
//       the 32 fp32 MFMAs they would replace, at four waves per SIMD
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

static inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t f2bf_trunc(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

// one wave per problem: A[16][16], B[16][16] bf16, C[16][16] f32 -> D
__global__ void k_sem(const uint16_t* A, const uint16_t* B, const float* C, float* D, int n) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n) return;
    const int r = lane & 15, kg = lane >> 4;
    const uint16_t* a = A + (size_t)t * 256;
    const uint16_t* b = B + (size_t)t * 256;
    bf16x4 av, bv;
    for (int i = 0; i < 4; ++i) { av[i] = (short)a[r * 16 + 4 * kg + i]; bv[i] = (short)b[(4 * kg + i) * 16 + r]; }
    f32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = C[(size_t)t * 256 + (4 * kg + i) * 16 + r];
    acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av, bv, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(size_t)t * 256 + (4 * kg + i) * 16 + r] = acc[i];
}

__global__ void k_rate(float* out, int iters, int mode, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    if (mode == 0) {  // what the kernels do today: 8 k-steps x 4 column tiles of v_mfma_f32_16x16x4_f32 per 16-row tile
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const float a = 1.0f + lane * 1e-3f, b = 0.5f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {  // mode = number of split terms (6 or 9): terms x 2 k-halves x 4 column tiles of v_mfma_f32_16x16x16_bf16
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        bf16x4 a = {(short)0x3f80, (short)0x3f00, (short)0x3e80, (short)(0x3e00 + lane)}, b = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80};
        for (int it = 0; it < iters; ++it)
            for (int term = 0; term < mode; ++term)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, acc[c], 0, 0, 0);
        r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
    const int n = 4096;
    std::vector<uint16_t> A(n * 256), B(n * 256);
    std::vector<float> C(n * 256), D(n * 256);
    srand(7);
    auto rnd = [&]() { return (float)rand() / RAND_MAX; };
    for (int t = 0; t < n; ++t)
        for (int i = 0; i < 256; ++i) {
            // wide exponent range and mixed signs (cancellation), like activations x weights
            const float sa = (rand() & 1) ? 1.f : -1.f, sb = (rand() & 1) ? 1.f : -1.f;
            A[t * 256 + i] = f2bf_trunc(sa * ldexpf(0.5f + rnd(), (rand() % 24) - 12));
            B[t * 256 + i] = f2bf_trunc(sb * ldexpf(0.5f + rnd(), (rand() % 24) - 12));
            C[t * 256 + i] = (t & 1) ? 0.f : ((rand() & 1) ? 1.f : -1.f) * ldexpf(0.5f + rnd(), (rand() % 24) - 12);
        }
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sem, dim3(n / 4), dim3(256), 0, 0, dA, dB, dC, dD, n);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    long total = 0, m_seq = 0, m_seq_c_last = 0, m_exact = 0, m_grp_c_first = 0, m_grp_c_last = 0, m_tree = 0, m_grp_tree = 0;
    for (int t = 0; t < n; ++t)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const float c = C[t * 256 + i * 16 + j], d = D[t * 256 + i * 16 + j];
                float p[16];
                double pd[16];
                for (int k = 0; k < 16; ++k) {
                    p[k] = bf2f(A[t * 256 + i * 16 + k]) * bf2f(B[t * 256 + k * 16 + j]);  // exact in float32
                    pd[k] = (double)bf2f(A[t * 256 + i * 16 + k]) * (double)bf2f(B[t * 256 + k * 16 + j]);
                }
                float s1 = c; for (int k = 0; k < 16; ++k) s1 = s1 + p[k];                  // sequential float32 chain from C
                float s1b = 0.f; for (int k = 0; k < 16; ++k) s1b = s1b + p[k]; s1b = s1b + c;  // ... C last
                long double e = (long double)c; for (int k = 0; k < 16; ++k) e += (long double)pd[k];
                const float s2 = (float)e;                                                // exact sum, one rounding
                float s3 = c, s3b = 0.f;                                                  // exact 4-term group sums, groups in order
                float gs[4];
                for (int g = 0; g < 4; ++g) { long double q = 0; for (int k = 0; k < 4; ++k) q += (long double)pd[4 * g + k]; gs[g] = (float)q; }
                for (int g = 0; g < 4; ++g) { s3 = s3 + gs[g]; s3b = s3b + gs[g]; }
                s3b = s3b + c;
                const float s4 = ((((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]))) +
                                  (((p[8] + p[9]) + (p[10] + p[11])) + ((p[12] + p[13]) + (p[14] + p[15])))) + c;  // pairwise tree
                const float s5 = ((gs[0] + gs[1]) + (gs[2] + gs[3])) + c;
                ++total;
                m_seq += (s1 == d); m_seq_c_last += (s1b == d); m_exact += (s2 == d); m_grp_c_first += (s3 == d);
                m_grp_c_last += (s3b == d); m_tree += (s4 == d); m_grp_tree += (s5 == d);
            }
    printf("v_mfma_f32_16x16x16_bf16 semantics, %ld outputs (wide exponents, cancellation, half with C = 0):\n", total);
    printf("  == sequential f32 chain from C            : %ld\n", m_seq);
    printf("  == sequential f32 chain, C added last      : %ld\n", m_seq_c_last);
    printf("  == exact sum of 16 products + C, rounded once: %ld\n", m_exact);
    printf("  == exact 4-term group sums, C + g0 + g1 + ..: %ld\n", m_grp_c_first);
    printf("  == exact 4-term group sums, ... + C last    : %ld\n", m_grp_c_last);
    printf("  == pairwise f32 tree + C                    : %ld\n", m_tree);
    printf("  == (g0 + g1) + (g2 + g3) + C                : %ld\n", m_grp_tree);
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4 * 2); hipMalloc(&cyc, 8 * 1024);
    for (int mode : {0, 6, 9}) {
        const int iters = 2000;
        hipLaunchKernelGGL(k_rate, dim3(256), dim3(1024), 0, 0, out, iters, mode, cyc);  // 16 waves per CU = 4 per SIMD
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
        printf("rate: %s per 16-row tile: %.1f clock ticks of s_memtime per tile and wave (4 waves per SIMD)\n",
               mode == 0 ? "32 x v_mfma_f32_16x16x4_f32" : mode == 6 ? "48 x v_mfma_f32_16x16x16_bf16 (6 terms)" : "72 x v_mfma_f32_16x16x16_bf16 (9 terms)",
               avg / iters);
    }
    return 0;
}
