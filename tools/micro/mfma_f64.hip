// Micro-test: is v_mfma_f64_16x16x4_f64 on gfx950 an exactly reproducible chain of IEEE double FMAs in ascending k
// (D[i][j] = fma(A[i][3], B[3][j], fma(A[i][2], B[2][j], fma(A[i][1], B[1][j], fma(A[i][0], B[0][j], C[i][j]))))?
// And what do the candidates for a double-precision 32-term dot product cost: the f64 MFMA, v_fma_f64 on the VALU,
// v_cvt_f64_f32?  (the transform of layer 1 is carried in double.)
// Inputs are float32 values widened to double with strong cancellation (as in that layer), 8 chained MFMAs = k 0..31.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));

// one wave per tile: H[16][32] f32, W[32][16] f32 -> Z[16][16] = float(double chain)
__global__ void k_tile(const float* H, const float* W, float* Z, double* Zd, int tiles) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= tiles) return;
    const int r = lane & 15, kq = lane >> 4;
    const float* h = H + (size_t)t * 16 * 32;
    const float* w = W + (size_t)t * 32 * 16;
    f64x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const double a = (double)h[r * 32 + 4 * s + kq];   // A[i = r][k = kq]
        const double b = (double)w[(4 * s + kq) * 16 + r]; // B[k = kq][j = r]
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // raw dump: [lane][reg]; the host works out which (row, col) that is
        Zd[(size_t)t * 256 + lane * 4 + i] = acc[i];
        Z[(size_t)t * 256 + lane * 4 + i] = (float)acc[i];
    }
}

__global__ void k_rate(double* out, int iters, int mode, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double r = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {  // f64 MFMA, 4 independent accumulators
        f64x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const double a = 1.0 + lane * 1e-3, b = 0.5;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else if (mode == 1) {  // v_fma_f64, 16 independent accumulators
        double acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = i;
        const double a = 1.0 + lane * 1e-9, b = 1e-3;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = fma(a, acc[i], b);
        for (int i = 0; i < 16; ++i) r += acc[i];
    } else if (mode == 2) {  // v_cvt_f64_f32 + v_fma_f64 pairs
        double acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = i;
        float x = 1.0f + lane * 1e-6f;
        const double b = 1e-3;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float xi = x + (float)i;
                asm volatile("" : "+v"(xi));
                acc[i] = fma((double)xi, acc[i], b);
            }
            x += 1e-7f;
        }
        for (int i = 0; i < 16; ++i) r += acc[i];
    } else {  // f32 MFMA 16x16x4 for reference
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        const float a = 1.0f + lane * 1e-3f, b = 0.5f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
        r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

int main() {
    const int tiles = 20000;
    std::vector<float> H((size_t)tiles * 512), W((size_t)tiles * 512);
    srand(12345);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (int t = 0; t < tiles; ++t) {
        const float scale = powf(10.f, (float)(t % 7) - 3.f);  // mixed magnitudes
        for (int i = 0; i < 512; ++i) { H[(size_t)t * 512 + i] = rnd() * scale; W[(size_t)t * 512 + i] = rnd(); }
        if (t & 1)  // strong cancellation: second half of k mirrors the first with opposite sign, slightly perturbed
            for (int r = 0; r < 16; ++r)
                for (int k = 16; k < 32; ++k) H[(size_t)t * 512 + r * 32 + k] = -H[(size_t)t * 512 + r * 32 + k - 16] * (1.f + 1e-6f * rnd());
        if (t & 2)
            for (int k = 16; k < 32; ++k)
                for (int c = 0; c < 16; ++c) W[(size_t)t * 512 + k * 16 + c] = W[(size_t)t * 512 + (k - 16) * 16 + c];
    }
    float *dH, *dW, *dZ; double* dZd;
    (void)hipMalloc(&dH, H.size() * 4); (void)hipMalloc(&dW, W.size() * 4); (void)hipMalloc(&dZ, (size_t)tiles * 256 * 4);
    (void)hipMalloc(&dZd, (size_t)tiles * 256 * 8);
    (void)hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_tile, dim3((tiles + 3) / 4), dim3(256), 0, 0, dH, dW, dZ, dZd, tiles);
    (void)hipDeviceSynchronize();
    std::vector<float> Z((size_t)tiles * 256); std::vector<double> Zd((size_t)tiles * 256);
    (void)hipMemcpy(Z.data(), dZ, Z.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(Zd.data(), dZd, Zd.size() * 8, hipMemcpyDeviceToHost);
    // candidate output layouts: lane l (r = l & 15, q = l >> 4), register i
    const char* lay_name[] = {"row 4q+i, col r", "row 4i+q, col r", "row r, col 4q+i", "row r, col 4i+q"};
    for (int lay = 0; lay < 4; ++lay) {
        size_t bad_chain = 0, bad_chain32 = 0, bad_rev = 0, bad_pair = 0;
        for (int t = 0; t < tiles; ++t)
            for (int l = 0; l < 64; ++l)
                for (int ri = 0; ri < 4; ++ri) {
                    const int r = l & 15, q = l >> 4;
                    const int i = lay == 0 ? 4 * q + ri : lay == 1 ? 4 * ri + q : r;
                    const int j = lay == 0 ? r : lay == 1 ? r : lay == 2 ? 4 * q + ri : 4 * ri + q;
                    const float* h = &H[(size_t)t * 512 + i * 32];
                    const float* w = &W[(size_t)t * 512 + j];
                    double c = 0, rv = 0, pr = 0;
                    for (int k = 0; k < 32; ++k) c = fma((double)h[k], (double)w[k * 16], c);
                    for (int s = 0; s < 8; ++s) {  // alternatives: descending k inside a block of 4; pairwise inside a block
                        for (int k = 3; k >= 0; --k) rv = fma((double)h[4 * s + k], (double)w[(4 * s + k) * 16], rv);
                        const double p01 = fma((double)h[4 * s], (double)w[(4 * s) * 16], (double)h[4 * s + 1] * (double)w[(4 * s + 1) * 16]);
                        const double p23 = fma((double)h[4 * s + 2], (double)w[(4 * s + 2) * 16], (double)h[4 * s + 3] * (double)w[(4 * s + 3) * 16]);
                        pr = pr + (p01 + p23);
                    }
                    const double g = Zd[(size_t)t * 256 + l * 4 + ri];
                    bad_chain += g != c;
                    bad_rev += g != rv;
                    bad_pair += g != pr;
                    bad_chain32 += Z[(size_t)t * 256 + l * 4 + ri] != (float)c;
                }
        printf("layout [%s]: of %zu outputs, vs ascending fma chain %zu differ (as float32: %zu); descending-in-block %zu; pairwise-in-block %zu\n",
               lay_name[lay], (size_t)tiles * 256, bad_chain, bad_chain32, bad_rev, bad_pair);
    }
    double* dout; unsigned long long* dcyc;
    (void)hipMalloc(&dout, 256 * 1024 * 8); (void)hipMalloc(&dcyc, 256 * 16 * 8);
    const char* names[] = {"v_mfma_f64_16x16x4_f64", "v_fma_f64", "v_cvt_f64_f32 + v_fma_f64", "v_mfma_f32_16x16x4_f32"};
    const double per_iter[] = {32, 32, 16, 32};
    for (int waves = 4; waves <= 16; waves *= 2)
        for (int m = 0; m < 4; ++m) {
            const int iters = 1000;
            hipLaunchKernelGGL(k_rate, dim3(256), dim3(64 * waves), 0, 0, dout, iters, m, dcyc);
            (void)hipDeviceSynchronize();
            std::vector<unsigned long long> c(256 * 16);
            (void)hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
            double a = 0;
            for (int g = 0; g < 256; ++g) for (int w = 0; w < waves; ++w) a += (double)c[g * 16 + w];
            a /= 256.0 * waves;
            // (s_memtime counts shader cycles on this part: v_mfma_f32_16x16x4_f32 reads 32.0 with one wave per SIMD)
            printf("%2d waves/CU  %-28s %9.0f cycles  -> %.2f cycles per instruction per wave, %.2f per SIMD-instruction\n",
                   waves, names[m], a, a / (iters * per_iter[m]), a / (iters * per_iter[m]) / (waves / 4.0));
        }
    return 0;
}
