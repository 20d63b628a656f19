// One GraphConvolution layer AND the next layer's feature transform in a single launch: the layer-by-layer path
// (mode 0: shapes the whole-path kernel of fused.hip does not take, and every caller that asks for it) at hidden width 32.
//
//     H' = act(Z0 + L . Z1 + b)            gcn/layers.py:206-216 of layer l      (what k_spmm_lds does)
//     Z' = H' . [W0' | W1']                gcn/layers.py:202-203 of layer l + 1  (what k_transform_* does)
//
// Separately that is two launches per layer and H' makes a round trip through HBM (12.8 MB written, 12.8 MB read
// per C3 layer: 93.5 MB of traffic in all); fused, H' only ever exists in LDS: 67.9 MB per layer and one launch.
// One 1024-thread workgroup per graph: stage the graph's Z1 slice and its (col, val) pairs in LDS, aggregate 8 rows
// per wave (8 lanes x float4 per row, sequential fmaf chain in CSR order - the library's arithmetic contract), park
// the activated rows in LDS over the dead Z1 slice, then run the 32 -> 64 product on fp32 MFMA 32x32x2 tiles
// (k-ordered fmaf chain, bit-identical to k_transform_mfma) or, when the next layer is the 32 -> 1 output layer, as
// two 32-term chains per vertex (bit-identical to k_transform_narrow).
#include "common.h"

namespace dgcn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
struct __attribute__((aligned(8))) ColValL { int col; float val; };

constexpr int kLC = 32;        // hidden width handled here
constexpr int kLHs = kLC + 1;  // LDS row stride of the parked H' (conflict-free column reads for the MFMA A operand)

__device__ __forceinline__ float4 lfma(float a, float4 z, float4 acc) {
    acc.x = fmaf(a, z.x, acc.x); acc.y = fmaf(a, z.y, acc.y); acc.z = fmaf(a, z.z, acc.z); acc.w = fmaf(a, z.w, acc.w);
    return acc;
}

// (8 waves per SIMD = 64 VGPRs: two 1024-thread workgroups per CU, one staging while the other computes)
template <int CTN>  // columns of the next layer's [W0' | W1']: 64 (hidden layer) or 2 (output layer)
__global__ __launch_bounds__(1024, 8) void k_layer32(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                  const float* __restrict__ values, const int32_t* __restrict__ graph_ptr,
                                                  const float* __restrict__ Z, const float* __restrict__ bias, int act,
                                                  const float* __restrict__ Wn, float* __restrict__ Zn, int csr_cap,
                                                  int zs_rows) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = blockIdx.x;
    const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1];
    const int ng = n1 - n0;
    if (ng <= 0) return;
    constexpr int ldz = 2 * kLC;
    float* zsm = smem;                                                          // [ng][32] Z1 slice
    ColValL* cvs = reinterpret_cast<ColValL*>(smem + (size_t)zs_rows * kLC);  // [csr_cap]
    const int e0 = row_ptr[n0], e1 = row_ptr[n1];
    const bool csr_in_lds = (e1 - e0) <= csr_cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, idx = lane & 31;
    // ---- stage the Z1 slice (coalesced 16-byte loads, 4 in flight per thread) and the (col, val) pairs.  Loads past the
    // end are clamped to the last element instead of predicated: branch-free, so the four loads really are in flight
    // together (a predicated version kept its temporaries in scratch memory and waited for each load in turn).
    {
        const int total = ng * (kLC / 4);
        for (int base = threadIdx.x; base < total; base += 1024 * 4) {
            float4 tmp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(base + u * 1024, total - 1);
                tmp[u] = *reinterpret_cast<const float4*>(Z + (size_t)(n0 + (i >> 3)) * ldz + kLC + (i & 7) * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * 1024;
                if (i < total) *reinterpret_cast<float4*>(zsm + (i >> 3) * kLC + (i & 7) * 4) = tmp[u];
            }
        }
    }
    if (csr_in_lds) {
        const int total = e1 - e0;
        for (int base = threadIdx.x; base < total; base += 1024 * 4) {
            int tc[4];
            float tv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(base + u * 1024, total - 1);
                tc[u] = col_idx[e0 + i];
                tv[u] = values[e0 + i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = base + u * 1024;
                if (i < total) { ColValL m; m.col = (tc[u] - n0) * kLC; m.val = tv[u]; cvs[i] = m; }
            }
        }
    }
    // ---- row bounds and the "+ Z0" operand of this thread's rows, fetched beside the staging loads
    constexpr int kSlots = 16 * 8;  // rows per pass: 16 waves x 8 rows
    constexpr int kPre = 4;         // passes: graphs up to 512 vertices
    const int sub = lane & 7, slot = wave * 8 + (lane >> 3);
    const int foff = sub * 4;
    int rs[kPre], re[kPre];
    float4 outv[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        const int v = slot + k * kSlots;
        rs[k] = re[k] = 0;
        outv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v < ng) {
            rs[k] = row_ptr[n0 + v];
            re[k] = row_ptr[n0 + v + 1];
            outv[k] = *reinterpret_cast<const float4*>(Z + (size_t)(n0 + v) * ldz + foff);  // Z0
        }
    }
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv = *reinterpret_cast<const float4*>(bias + foff);
    __syncthreads();
    // ---- aggregate: sequential fmaf chain over the row's entries in CSR order from 0, then Z0 + sum, + bias, activation
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        if (wave * 8 + k * kSlots >= ng) continue;  // wave-uniform (no break: the unrolled loop keeps rs/re/outv in registers)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int j = rs[k];
        if (csr_in_lds) {
            for (; j + 1 < re[k]; j += 2) {
                const ColValL m0 = cvs[j - e0], m1 = cvs[j + 1 - e0];
                const float4 z0 = *reinterpret_cast<const float4*>(zsm + m0.col + foff);
                const float4 z1 = *reinterpret_cast<const float4*>(zsm + m1.col + foff);
                acc = lfma(m0.val, z0, acc);
                acc = lfma(m1.val, z1, acc);
            }
            if (j < re[k]) {
                const ColValL m = cvs[j - e0];
                acc = lfma(m.val, *reinterpret_cast<const float4*>(zsm + m.col + foff), acc);
            }
        } else {
            for (; j < re[k]; ++j)
                acc = lfma(values[j], *reinterpret_cast<const float4*>(zsm + (col_idx[j] - n0) * kLC + foff), acc);
        }
        float4 o = outv[k];
        o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;  // Z0 + sum
        if (bias) { o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
        outv[k] = make_float4(apply_act(o.x, act), apply_act(o.y, act), apply_act(o.z, act), apply_act(o.w, act));
    }
    __syncthreads();  // every gather is done: the Z1 slice and the pairs are dead
    // ---- park H' in LDS (row stride 33)
    float* hs = smem;
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        const int v = slot + k * kSlots;
        if (v < ng) {
            float* dst = hs + v * kLHs + foff;
            dst[0] = outv[k].x; dst[1] = outv[k].y; dst[2] = outv[k].z; dst[3] = outv[k].w;
        }
    }
    __syncthreads();
    // ---- next layer's transform: one (32-row tile, 32-column half) unit per wave at a time, so a wave holds the B
    // fragments of ONE column half (16 registers): lane holds W'[2s + half][ct*32 + idx]
    if (CTN == 64) {
        const int units = ((ng + 31) >> 5) * 2;
        for (int u = wave; u < units; u += 16) {
            const int row0 = (u >> 1) * 32, ct = u & 1;
            float bfrag[16], afrag[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) bfrag[s] = Wn[(2 * s + half) * CTN + ct * 32 + idx];
#pragma unroll
            for (int s = 0; s < 16; ++s) afrag[s] = hs[(row0 + idx) * kLHs + 2 * s + half];  // rows past ng: unused garbage
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afrag[s], bfrag[s], acc, 0, 0, 0);
            // C/D map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
                if (r < ng) Zn[(size_t)(n0 + r) * CTN + ct * 32 + idx] = acc[reg];
            }
        }
    } else {
        for (int v = threadIdx.x; v < ng; v += 1024) {
            float z0 = 0.f, z1 = 0.f;
#pragma unroll
            for (int k = 0; k < kLC; ++k) {
                const float h = hs[v * kLHs + k];
                z0 = fmaf(h, Wn[k * 2 + 0], z0);
                z1 = fmaf(h, Wn[k * 2 + 1], z1);
            }
            Zn[(size_t)(n0 + v) * 2 + 0] = z0;
            Zn[(size_t)(n0 + v) * 2 + 1] = z1;
        }
    }
}

int spmm_split_for(int C);

// 1 = launched, 0 = this shape is not handled here (the caller runs the two separate kernels), < 0 = error.
int layer32_dispatch(const DgcnCsr* S, const int32_t* graph_ptr, int B, int max_nodes, const float* Z, const float* bias, int act,
                     const float* Wn, int ctot_next, float* Zn, hipStream_t s) {
    if (!graph_ptr || B <= 0 || max_nodes <= 0 || max_nodes > 512) return 0;
    if (ctot_next != 64 && ctot_next != 2) return 0;
    if (spmm_split_for(kLC) != 1) return 0;  // a tuning override of the summation order: keep the plain kernels
    if (opt(OPT_LAYER_FUSE) == 0) return 0;
    if (((uintptr_t)Z | (uintptr_t)Zn | (uintptr_t)(bias ? bias : Z)) % 16) return 0;
    constexpr size_t kLdsMax = 150 * 1024;
    const int rows32 = (max_nodes + 31) & ~31;
    const size_t zbytes = (size_t)max_nodes * kLC * sizeof(float);
    const size_t hbytes = (size_t)rows32 * kLHs * sizeof(float);
    if (hbytes + 1024 > kLdsMax) return 0;
    long cap = S->max_graph_nnz > 0 ? S->max_graph_nnz : (long)((double)S->nnz / max(S->num_rows, 1) * max_nodes * 1.15) + 128;
    const long two_per_cu = ((long)78 * 1024 - (long)zbytes) / 8;  // prefer two workgroups per CU when the pairs still fit
    (void)two_per_cu;
    cap = min(cap, (long)((kLdsMax - zbytes) / 8));
    const int csr_cap = (int)(cap & ~3L);
    const size_t lds = max(zbytes + (size_t)csr_cap * 8, hbytes);
    hipError_t e = hipSuccess;
    if (lds > 64 * 1024) {
        e = ctot_next == 64 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&k_layer32<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                            : hipFuncSetAttribute(reinterpret_cast<const void*>(&k_layer32<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DGCN_ERR_LAUNCH, "k_layer32: cannot reserve %zu bytes of LDS", lds);
    }
    TimedLaunch t("layer", s);
    if (ctot_next == 64)
        DGCN_LAUNCH(t, (k_layer32<64>), dim3(B), dim3(1024), lds, s, S->row_ptr, S->col_idx, S->values, graph_ptr, Z, bias, act, Wn, Zn,
                    csr_cap, max_nodes);
    else
        DGCN_LAUNCH(t, (k_layer32<2>), dim3(B), dim3(1024), lds, s, S->row_ptr, S->col_idx, S->values, graph_ptr, Z, bias, act, Wn, Zn,
                    csr_cap, max_nodes);
    int rc = check_launch("k_layer32");
    return rc ? rc : 1;
}

}  // namespace dgcn
