// Tile-level pieces of the in-LDS GraphConvolution stack, shared by fused.hip (k_fused) and tail.hip (k_tail): the swizzled
// [N][32] image buffers, the weight fragments of a hidden layer and the two MFMA transforms.  Arithmetic contract as in
// include/dgcn.h ("Precision"): every output element is the k-ordered fmaf chain (fp32 MFMA 16x16x4 is exactly that on gfx950),
// layer index 1 the ascending fma chain in double, rounded once (v_mfma_f64_16x16x4_f64).
#pragma once
#include "common.h"

namespace dgcn {

constexpr int kHid = 32;  // hidden width of the LDS image

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ int swz(int row, int col) {  // float index of H[row][col] in a swizzled buffer
    return row * kHid + ((((col >> 2) ^ (row & 7)) << 2) | (col & 3));
}
// (Tried and dropped: rows of bufA / bufB in a permuted slot order - slot p = feature 4 * (p % 8) + p / 8 - that makes the
// eight features 4s + kq of an MFMA lane two ds_read_b128 instead of eight ds_read_b32.  The reads got 2 us cheaper per
// launch, but every wave then fetches its weight fragments with a stride of four columns and the doubled L1 traffic of
// that cost 30 us.)

// bufB swizzle key of a row (xor-ed into the chunk index, inside the 64-byte half)
__device__ __forceinline__ int keyB(int row) { return (row >> 1) & 3; }
__device__ __forceinline__ unsigned short enc_word(int u) { return (unsigned short)((u << 7) | (keyB(u) << 4)); }
__device__ __forceinline__ int swzB(int row, int col) {
    return row * kHid + ((((col >> 2) ^ keyB(row)) << 2) | (col & 3));
}

// ---- hidden layer 32 -> (32 | 32): fp32 MFMA 16x16x4, one 16-row tile per wave at a time.
// A: lane (r = l & 15, kq = l >> 4) holds H[row0 + r][4s + kq]; B: W[4s + kq][ct*16 + r];
// C/D: col = l & 15, row = 4 * (l >> 4) + reg.  Z0 overwrites the tile's own rows of bufA.
// The B fragments of a layer are fetched one layer ahead (load_bfrag) so that their global-memory
// latency hides under the previous layer's gather phase.
// `f64map`: fragments for hidden_transform_f64 - the f64 MFMA returns rows 4 * reg + (lane >> 4) where the f32 one
// returns 4 * (lane >> 4) + reg, so lane r feeds column 4 * (r & 3) + (r >> 2) of the tile and the accumulator again
// holds four CONSECUTIVE features per lane.
__device__ __forceinline__ void load_bfrag(const float* W, float (&b)[8][4], int skip = 0, bool f64map = false) {
    const int lane = threadIdx.x & 63;
#ifdef DGCN_DIAG
    if (skip) {  // experiment: no weight fetch
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) b[s][ct] = 0.01f * (float)(s + ct);
        return;
    }
#endif
    (void)skip;
    const int r0 = lane & 15, kq = lane >> 4;
    const int r = f64map ? 4 * (r0 & 3) + (r0 >> 2) : r0;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) b[s][ct] = W[(4 * s + kq) * 64 + ct * 16 + r];
}

// (Tried and dropped: sending the 1..8 rows a vertex count leaves over - ER N = 200: 12 full tiles + 8 rows, which puts a
// 4th tile on one SIMD - through the VALU of the last wave instead of a 13th MFMA tile (lane = output column, 32-term
// fmaf chain per row).  Same bits, but 252 us instead of 219 us per C3 launch: the wave's 32 column weights spill at
// the 128-VGPR budget and its serial rows become the phase's critical path.)
template <int BLOCK>
__device__ __forceinline__ void hidden_transform(const float (&b)[8][4], int ng, float* bufA, float* bufB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    constexpr int kWaves = BLOCK / 64;
    const int tiles = (ng + 15) >> 4;
    for (int t = wave; t < tiles; t += kWaves) {
        const int row = t * 16 + r;
        float av[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) av[s] = bufA[row * kHid + (((s ^ (row & 7)) << 2) | kq)];
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // Operands swapped (D^T = W^T . H^T): the accumulator then holds, per lane, 4 CONSECUTIVE
        // features (4*kq + reg of column tile ct) of ONE vertex (row0 + r) = one 16-byte chunk,
        // stored with a single ds_write_b128.  Each element is still the k-ordered fmaf chain.
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s][ct], av[s], acc[ct], 0, 0, 0);
        if (row < ng) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int chunk = (ct & 1) * 4 + kq;
                const float4 o = make_float4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]);
                if (ct < 2) *reinterpret_cast<float4*>(bufA + row * kHid + ((chunk ^ (row & 7)) << 2)) = o;
                else *reinterpret_cast<float4*>(bufB + row * kHid + ((chunk ^ keyB(row)) << 2)) = o;
            }
        }
    }
}

// The same product with every chain carried in double and rounded once (layer index 1): v_mfma_f64_16x16x4_f64, which
// on gfx950 is exactly fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, c)))) per element (tools/micro/mfma_f64.hip: 0 of
// 5.1 M outputs differ from the CPU's chain, cancellation included).  Fragments from load_bfrag(.., f64map = true).
// Column tiles in pairs: 16 accumulator registers live at a time.
using f64x4 = __attribute__((ext_vector_type(4))) double;
template <int BLOCK>
__device__ __forceinline__ void hidden_transform_f64(const float (&b)[8][4], int ng, float* bufA, float* bufB) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    constexpr int kWaves = BLOCK / 64;
    const int tiles = (ng + 15) >> 4;
    for (int t = wave; t < tiles; t += kWaves) {
        const int row = t * 16 + r;
        float av[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) av[s] = bufA[row * kHid + (((s ^ (row & 7)) << 2) | kq)];
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
            f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                // (opaque copies: otherwise the 32 conversions of b are hoisted out of the tile loop and their 64 registers
                // push the row-block state of the aggregation into scratch memory)
                float b0 = b[s][2 * cp], b1 = b[s][2 * cp + 1], a0 = av[s];
                asm volatile("" : "+v"(b0), "+v"(b1), "+v"(a0));
                const double ad = (double)a0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b0, ad, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b1, ad, acc1, 0, 0, 0);
            }
            if (row < ng) {
                const float4 o0 = make_float4((float)acc0[0], (float)acc0[1], (float)acc0[2], (float)acc0[3]);
                const float4 o1 = make_float4((float)acc1[0], (float)acc1[1], (float)acc1[2], (float)acc1[3]);
                float* dst = cp == 0 ? bufA : bufB;
                const int key = cp == 0 ? (row & 7) : keyB(row);
                *reinterpret_cast<float4*>(dst + row * kHid + ((kq ^ key) << 2)) = o0;
                *reinterpret_cast<float4*>(dst + row * kHid + (((4 + kq) ^ key) << 2)) = o1;
            }
        }
    }
}

__device__ __forceinline__ float4 fma4(float a, float4 z, float4 acc) {
    acc.x = fmaf(a, z.x, acc.x); acc.y = fmaf(a, z.y, acc.y); acc.z = fmaf(a, z.z, acc.z); acc.w = fmaf(a, z.w, acc.w);
    return acc;
}

}  // namespace dgcn
