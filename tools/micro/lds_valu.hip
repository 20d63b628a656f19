// Micro-benchmark: does VALU work slow the LDS gather stream down on gfx950?  16 waves per CU (one 1024-thread
// workgroup), every wave loops over trips of 8 ds_read_b128 (4 lanes per row, half by slot, random rows - the
// aggregation's pattern) followed by NF packed FMAs that consume the gathered data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NF>
__global__ void k(const unsigned* __restrict__ idx, float* out, int iters, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 128 * 32; i += blockDim.x) lds[i] = (float)(i & 1023) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int s = lane >> 2;
    const unsigned off = (unsigned)((lane & 3) | (((s >> 1) & 1) << 2)) << 4;
    unsigned ad[8];
    const int wid = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    for (int i = 0; i < 8; ++i) ad[i] = (idx[(wid * 16 * 8 + s * 8 + i) & ((1 << 20) - 1)] & 127u) * 128u + off;
    f32x2 acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    const f32x2 a = {1.0f + lane * 1e-3f, 0.5f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned flip = (unsigned)(it & 63) << 8;
        f32x4 z[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(z[i]) : "v"(ad[i] ^ flip));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const f32x4 zz = z[f & 7];
            const f32x2 lo = {zz[0], zz[1]}, hi = {zz[2], zz[3]};
            acc[(2 * f) & 3] = __builtin_elementwise_fma(a, (f & 8) ? hi : lo, acc[(2 * f) & 3]);
            acc[(2 * f + 1) & 3] = __builtin_elementwise_fma(a, (f & 8) ? lo : hi, acc[(2 * f + 1) & 3]);
        }
        if (NF == 0) acc[0][0] += z[0][0] + z[1][1] + z[2][2] + z[3][3] + z[4][0] + z[5][1] + z[6][2] + z[7][3];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][0] + acc[3][1];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NF>
void run(unsigned* didx, float* dout, unsigned long long* dcyc) {
    const int iters = 2000, blocks = 256;
    hipLaunchKernelGGL(k<NF>, dim3(blocks), dim3(1024), 128 * 128, 0, didx, dout, iters, dcyc);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> c(blocks);
    (void)hipMemcpy(c.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto x : c) avg += x;
    avg /= blocks;
    printf("8 gathers + %2d packed FMAs per trip: %7.1f cycles per trip of one wave = %5.2f CU-cycles per gather; VALU alone would need %5.1f cycles per trip round of a SIMD\n",
           2 * NF, avg / iters, avg / (iters * 8.0 * 16), 4.0 * (2 * NF + 8) * 4);
}

int main() {
    unsigned* didx; float* dout; unsigned long long* dcyc;
    std::vector<unsigned> h(1 << 20);
    unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s >> 8; }
    (void)hipMalloc(&didx, h.size() * 4); (void)hipMemcpy(didx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&dout, 1 << 24); (void)hipMalloc(&dcyc, 1 << 16);
    run<0>(didx, dout, dcyc);
    run<4>(didx, dout, dcyc);
    run<8>(didx, dout, dcyc);
    run<16>(didx, dout, dcyc);
    return 0;
}
