// Micro-benchmark: do fp32 MFMA and packed-fp32 VALU work of DIFFERENT waves overlap on a gfx950 SIMD?
// One 512-thread workgroup per CU (2 waves per SIMD).  Waves 0-3 run a chain of v_mfma_f32_16x16x4_f32 (4 independent
// accumulators), waves 4-7 a chain of v_pk_fma_f32 (8 independent accumulators), or LDS gathers.  mode bit0: MFMA waves
// work, bit1: VALU waves work, bit2: waves 4-7 do ds_read_b128 gathers instead of FMAs.  Prints cycles for each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void k(float* out, int iters, int mode, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)(i & 63) * 0.001f;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float r = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1) {
            f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const float a = 1.0f + lane * 1e-3f, b = 0.5f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
            }
            r = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        }
    } else if (mode & 4) {
        f32x4 acc = {0, 0, 0, 0};
        unsigned ad = (unsigned)((lane * 37) & 255) * 128u + (unsigned)(lane & 7) * 16u;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const f32x4 z = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(lds) + ((ad + i * 1024u) & 32767u));
                acc += z;
            }
        }
        r = acc[0] + acc[1] + acc[2] + acc[3];
    } else if (mode & 2) {
        f32x2 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x2){0.f, 0.f};
        const f32x2 a = {1.0f + lane * 1e-3f, 0.999f}, b = {1e-3f, 2e-3f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(a, acc[i], b);
        }
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* dout; unsigned long long* dcyc;
    (void)hipMalloc(&dout, 256 * 512 * 4); (void)hipMalloc(&dcyc, 256 * 8 * 8);
    const int iters = 2000;
    const int modes[] = {1, 2, 3, 4, 5};
    const char* names[] = {"MFMA waves alone", "pk_fma waves alone", "MFMA + pk_fma waves", "LDS-gather waves alone", "MFMA + LDS-gather waves"};
    for (int m = 0; m < 5; ++m) {
        hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, dout, iters, modes[m], dcyc);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> c(256 * 8);
        (void)hipMemcpy(c.data(), dcyc, c.size() * 8, hipMemcpyDeviceToHost);
        double a = 0, b = 0;
        for (int g = 0; g < 256; ++g) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += (double)c[g * 8 + w];
        a /= 1024; b /= 1024;
        printf("%-28s waves 0-3: %8.0f cycles (%.1f per MFMA)   waves 4-7: %8.0f cycles (%.2f per instr)\n", names[m], a, a / (iters * 32.0), b,
               b / (iters * ((modes[m] & 4) ? 32.0 : 64.0)));
    }
    return 0;
}
