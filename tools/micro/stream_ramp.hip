// What can ONE launch of W workgroups reach when every workgroup first reads its whole input (an ER N=200 graph-layer of the SpMM:
// 87 KB of CSR + Z), only then produces output (26 KB of Y) - with nothing but the memory traffic in it?  The ceiling of a
// 500-graph k_spmm_lds launch (18.5 us, 0.29 of the HBM peak on SURVEY 8d's bytes) against the same traffic as one 4 000-workgroup
// launch.  Buffers rotate so that every launch's lines were last touched > 320 MB ago (out of the 256 MiB Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 -o stream_ramp stream_ramp.hip && ./stream_ramp
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

constexpr int kBlock = 1024;
typedef float f4v __attribute__((ext_vector_type(4)));

// in_f4 / out_f4: float4s per workgroup.  Loads: four in flight per thread (k_spmm_lds's staged_copy), everything read before
// anything is written.  NT: nontemporal stores.
template <bool NT>
__global__ __launch_bounds__(kBlock) void k_move(const float4* __restrict__ in, float4* __restrict__ out, int in_f4, int out_f4) {
    const float4* src = in + (size_t)blockIdx.x * in_f4;
    float4* dst = out + (size_t)blockIdx.x * out_f4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = threadIdx.x; base < in_f4; base += kBlock * 4) {
        float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = base + u * kBlock < in_f4 ? src[base + u * kBlock] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
    }
    __shared__ float4 red[kBlock];
    red[threadIdx.x] = acc;
    __syncthreads();  // (like the SpMM: every output needs the whole input first)
    acc = red[(threadIdx.x * 7 + 3) & (kBlock - 1)];
    for (int i = threadIdx.x; i < out_f4; i += kBlock) {
        if (NT) { f4v q = {acc.x, acc.y, acc.z, acc.w}; __builtin_nontemporal_store(q, reinterpret_cast<f4v*>(dst + i)); }
        else dst[i] = acc;
    }
}

int main() {
    const int in_bytes = 87 * 1024, out_bytes = 26 * 1024;  // an ER N=200 p=0.1 graph-layer at C = 32 (85 372 B in + out incl. the row pointers)
    const int in_f4 = in_bytes / 16, out_f4 = out_bytes / 16;
    hipStream_t s;
    hipStreamCreate(&s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int W : {500, 1000, 2000, 4000}) {
        const size_t per_launch = (size_t)W * (in_bytes + out_bytes);
        const int nsets = (int)((size_t)360 * 1024 * 1024 / per_launch) + 2;
        std::vector<float4*> ins(nsets), outs(nsets);
        for (int i = 0; i < nsets; ++i) {
            hipMalloc(&ins[i], (size_t)W * in_bytes);
            hipMalloc(&outs[i], (size_t)W * out_bytes);
            hipMemsetAsync(ins[i], 0, (size_t)W * in_bytes, s);
        }
        hipStreamSynchronize(s);
        for (int nt = 0; nt < 2; ++nt) {
            double total_us = 0.0;
            int counted = 0;
            for (int rep = 0; rep < 6; ++rep)
                for (int i = 0; i < nsets; ++i) {
                    if (nt) hipExtLaunchKernelGGL(k_move<true>, dim3(W), dim3(kBlock), 0, s, e0, e1, 0, ins[i], outs[i], in_f4, out_f4);
                    else hipExtLaunchKernelGGL(k_move<false>, dim3(W), dim3(kBlock), 0, s, e0, e1, 0, ins[i], outs[i], in_f4, out_f4);
                    hipStreamSynchronize(s);
                    float ms = 0.f;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rep >= 2) { total_us += ms * 1e3; ++counted; }
                }
            const double us = total_us / counted;
            printf("%4d workgroups x (87 KB in, then 26 KB out), %d rotating buffer sets, %s stores: %.1f us per launch, %.0f GB/s (%.2f of 8 TB/s)\n",
                   W, nsets, nt ? "nontemporal" : "plain", us, per_launch / us / 1e3, per_launch / us / 1e3 / 8000.0);
        }
        for (int i = 0; i < nsets; ++i) { hipFree(ins[i]); hipFree(outs[i]); }
    }
    return 0;
}
