// Micro-benchmark (round 6): can the vector-memory path (TA / L1 / L2) carry part of the aggregation's gathers beside the LDS?
// k_fused's trip - 16 rows per wave, 4 lanes per row, 4 entries per trip, two 16-byte chunks per entry and lane: 8 ds_read_b128, then 16
// packed FMAs - with G of the 4 entries' chunks read from a copy of the rows in GLOBAL memory (25.6 KB per workgroup, L1 / L2 resident;
// global_load_dwordx4 with a scalar base) instead of from LDS.  Two 512-thread workgroups per CU (78 KB of LDS each), 200 rows.
//   hipcc --offload-arch=gfx950 -O3 -o mix_gather mix_gather.hip && ./mix_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int G>
__global__ __launch_bounds__(512) void k(const uint2* __restrict__ rec, const float* __restrict__ gtab, float* out, int trips, int nrows,
                                         unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const float* grows = gtab + (size_t)blockIdx.x * nrows * 32;
    for (int i = threadIdx.x; i < nrows * 32; i += blockDim.x) lds[i] = grows[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane >> 2, kq = lane & 3;
    const unsigned cA = (unsigned)(kq | (((s >> 1) & 1) << 2)) << 4, cB = cA ^ 64u;
    const char* base = reinterpret_cast<const char*>(lds);
    const char* gbase = reinterpret_cast<const char*>(grows);
    // records as k_fused reads them: one {value, word} per lane and trip, 512 consecutive bytes per wave, a trip ahead
    const char* bp = reinterpret_cast<const char*>(rec) + (size_t)((blockIdx.x & 63) * 8 + wave) * (size_t)(trips + 2) * 512;
    const unsigned voff = (unsigned)lane * 8u;
    float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = accA;
    uint2 cur = *reinterpret_cast<const uint2*>(bp + voff);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define QB(x, e) __builtin_amdgcn_update_dpp(0, (int)(x), (e) * 0x55, 0xf, 0xf, true)
    for (int t = 0; t < trips; ++t) {
        const uint2 nxt = *reinterpret_cast<const uint2*>(bp + (t + 1) * 512 + voff);
        float4 zA[4], zB[4];
        float av[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned w = (unsigned)(e == 0 ? QB(cur.y, 0) : e == 1 ? QB(cur.y, 1) : e == 2 ? QB(cur.y, 2) : QB(cur.y, 3));
            av[e] = __int_as_float(e == 0 ? QB(cur.x, 0) : e == 1 ? QB(cur.x, 1) : e == 2 ? QB(cur.x, 2) : QB(cur.x, 3));
            if (e >= 4 - G) {
                zA[e] = *reinterpret_cast<const float4*>(gbase + (w ^ cA));
                zB[e] = *reinterpret_cast<const float4*>(gbase + (w ^ cB));
            } else {
                zA[e] = *reinterpret_cast<const float4*>(base + (w ^ cA));
                zB[e] = *reinterpret_cast<const float4*>(base + (w ^ cB));
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            accA.x = fmaf(av[e], zA[e].x, accA.x); accA.y = fmaf(av[e], zA[e].y, accA.y); accA.z = fmaf(av[e], zA[e].z, accA.z); accA.w = fmaf(av[e], zA[e].w, accA.w);
            accB.x = fmaf(av[e], zB[e].x, accB.x); accB.y = fmaf(av[e], zB[e].y, accB.y); accB.z = fmaf(av[e], zB[e].z, accB.z); accB.w = fmaf(av[e], zB[e].w, accB.w);
        }
        cur = nxt;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = accA.x + accA.y + accA.z + accA.w + accB.x + accB.y + accB.z + accB.w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int G>
void run(uint2* didx, float* gtab, float* dout, unsigned long long* dcyc) {
    const int iters = 2000, nrows = 200, blocks = 512;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<G>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<G>, dim3(blocks), dim3(512), 78 * 1024, 0, didx, gtab, dout, iters, nrows, dcyc);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> c(blocks);
    hipMemcpy(c.data(), dcyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto x : c) avg += x;
    avg /= blocks;
    printf("entries per trip from global: %d of 4 -> %8.1f clock ticks (100 MHz) per 1000 trips per wave; %6.3f us per trip\n", G, avg / iters * 1000.0,
           avg / iters / 100.0);
}

int main() {
    uint2* didx; float* dout; float* gtab; unsigned long long* dcyc;
    const size_t nrec = (size_t)64 * 8 * (2000 + 2) * 64;  // 64 distinct workgroup streams (33 MB), re-used modulo 64: L2 / MALL resident like k_fused's
    std::vector<uint2> h(nrec);
    unsigned s = 12345;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x.y = ((s >> 8) % 200u) * 128u; x.x = 0x3f800000u | (s & 0xffu); }
    hipMalloc(&didx, nrec * 8); hipMemcpy(didx, h.data(), nrec * 8, hipMemcpyHostToDevice);
    hipMalloc(&gtab, (size_t)512 * 200 * 128); hipMemset(gtab, 0, (size_t)512 * 200 * 128);
    hipMalloc(&dout, 1 << 24); hipMalloc(&dcyc, 1 << 16);
    run<0>(didx, gtab, dout, dcyc);
    run<1>(didx, gtab, dout, dcyc);
    run<2>(didx, gtab, dout, dcyc);
    run<3>(didx, gtab, dout, dcyc);
    run<4>(didx, gtab, dout, dcyc);
    return 0;
}
