// Micro-benchmark: throughput of random-row gathers from LDS on gfx950, the inner operation of the
// GCN aggregation.  N rows of 32 floats in LDS; every lane group reads pseudo-random rows.
//   mode 0: ds_read_b128, 8 lanes per row   mode 1: ds_read_b64, 16 lanes per row
//   mode 2: ds_read_b32, 32 lanes per row   mode 3: b128, all groups read the SAME row (broadcast)
// Prints cycles per wave-instruction and bytes/clk/CU for 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void k(const unsigned* __restrict__ idx, float* out, int iters, int nrows, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < nrows * 32; i += blockDim.x) lds[i] = (float)(i & 1023);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const char* base = reinterpret_cast<const char*>(lds);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    constexpr int LPR = MODE == 1 ? 16 : (MODE == 2 ? 32 : 8);
    const int grp = lane / LPR, sub = lane % LPR;
    unsigned r[8];
    for (int i = 0; i < 8; ++i) r[i] = idx[((blockIdx.x * blockDim.x / 64 + (threadIdx.x >> 6)) * 64 * 8 + grp * 8 + i) & ((1 << 20) - 1)] % nrows;
    if (MODE == 3) for (int i = 0; i < 8; ++i) r[i] = idx[i] % nrows;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned row = (r[i] + it * 7) % (unsigned)nrows;
            if (MODE == 0 || MODE == 3) {
                const float4 z = *reinterpret_cast<const float4*>(base + row * 128 + sub * 16);
                acc0 += z.x; acc1 += z.y; acc2 += z.z; acc3 += z.w;
            } else if (MODE == 4) {   // metadata-style read: 8 lanes share one 8-byte pair (ds_read_b64, broadcast inside the group)
                const float2 z = *reinterpret_cast<const float2*>(base + row * 128 + (it & 15) * 8);
                acc0 += z.x; acc1 += z.y;
            } else if (MODE == 5) {   // ... and one 4-byte word pair
                const float z = *reinterpret_cast<const float*>(base + row * 128 + (it & 31) * 4);
                acc0 += z;
            } else if (MODE == 1) {
                const float2 z = *reinterpret_cast<const float2*>(base + row * 128 + sub * 8);
                acc0 += z.x; acc1 += z.y;
            } else {
                const float z = *reinterpret_cast<const float*>(base + row * 128 + sub * 4);
                acc0 += z;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, unsigned* didx, float* dout, unsigned long long* dcyc) {
    const int iters = 2000, nrows = 200;
    for (int waves = 4; waves <= 32; waves *= 2) {  // waves per CU, one workgroup per CU
        const int threads = min(1024, waves * 64);
        const int blocks_per_cu = (waves * 64) / threads;
        const int blocks = 256 * blocks_per_cu;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), nrows * 128, 0, didx, dout, iters, nrows, dcyc);
        hipDeviceSynchronize();
        std::vector<unsigned long long> c(blocks);
        hipMemcpy(c.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto x : c) avg += x;
        avg /= blocks;
        const double instr_per_wave = iters * 8.0;
        const double cyc_per_cu_instr = avg / (instr_per_wave * waves);  // CU-level cycles per wave-instruction
        const double bytes = (MODE == 1 || MODE == 4 ? 512.0 : (MODE == 2 || MODE == 5 ? 256.0 : 1024.0));
        printf("%-28s waves/CU %2d: %7.2f cycles per wave-instr per wave, %6.2f CU-cycles per instr, %6.1f B/clk/CU\n", name,
               waves, avg / instr_per_wave, cyc_per_cu_instr, bytes / cyc_per_cu_instr);
    }
}

int main() {
    unsigned* didx; float* dout; unsigned long long* dcyc;
    std::vector<unsigned> h(1 << 20);
    unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s >> 8; }
    hipMalloc(&didx, h.size() * 4); hipMemcpy(didx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dout, 1 << 24); hipMalloc(&dcyc, 1 << 16);
    run<0>("b128 x 8 lanes/row random", didx, dout, dcyc);
    run<1>("b64 x 16 lanes/row random", didx, dout, dcyc);
    run<2>("b32 x 32 lanes/row random", didx, dout, dcyc);
    run<3>("b128 same rows (broadcast)", didx, dout, dcyc);
    run<4>("b64 one address per 8 lanes", didx, dout, dcyc);
    run<5>("b32 one address per 8 lanes", didx, dout, dcyc);
    return 0;
}
