// The hidden layers of a deep GraphConvolution stack (gcn/layers.py:189-216, gcn/models.py:550-573: c -> c with
// c = 32) in ONE launch for graphs whose whole image does not fit a CU's LDS - the any-size path's forward (general.hip).
//
// fused.hip keeps H, Z0, Z1 and the support of a graph in LDS: 256 bytes per vertex + 6 per entry, <= 512 vertices.  A
// multi-channel joint conflict graph has K * nflows vertices (wireless_rollout_test_flood.py:98-133: 3 x 300), ER(500, 0.1)
// has 25 000 entries.  What an aggregation GATHERS at random is Z1 alone (128 bytes per vertex: 976 vertices fit); H and Z0
// are touched by the wave that owns their rows only, and the support is read front to back.  So here:
//
//   LDS        Z1[N][32] (bufB of fused.hip, same swizzle, same gather addresses), the row order, a zero row, a 2 KB
//              staging tile per wave
//   registers  H and Z0 of the (at most four) 16-row tiles a wave owns, in BOTH phases: Z0 in the aggregation's lane layout
//              from a transform to the next aggregation, H' in the MFMA's operand layout from an aggregation to the next
//              transform; converted through the wave's staging tile (no workgroup barrier: a wave's LDS operations complete
//              in order)
//   global     (L2 / MALL-resident scratch of the caller) the support as block-major padded 8-byte records {value, LDS
//              address of the neighbour's Z1 row} (fused.hip's row_blocks_init: 512 consecutive bytes per wave and trip),
//              written once per launch, requested four trips ahead
//   one 1 024-thread workgroup per graph; a wave owns the tiles t = wave, wave + 16, ... (rows in descending entry-count
//   order, so the 16 rows a wave walks in lockstep have similar lengths), two barriers per layer.
//
// The first layer (F -> 32, chains in double), the second layer's transform (double) and the last layer (32 -> 1) are the
// layer-by-layer kernels' (forward.hip): this kernel takes Z of layer index 1 and returns the last hidden activations.
// Arithmetic as everywhere (include/dgcn.h): transform = k-ordered fmaf chain (v_mfma_f32_16x16x4_f32), aggregation = fmaf
// chain over the row's entries in storage order from 0, then Z0 + sum, + bias, activation: bit-identical to mode 0.
//
// Bound: the LDS array (one 128-byte row of Z1 per entry and layer: entries x 19 x 128 B / (128 B/clk) per graph);
// HBM sees the support once per launch (the records are re-read from L2 / MALL by every layer).
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "common.h"

namespace dgcn {

constexpr int kBigBlock = 1024;
constexpr int kBigWaves = kBigBlock / 64;
constexpr int kBigMaxNodes = 976;   // Z1 (128 B per vertex) + a 2 KB staging tile per wave + the row tables in 160 KB; 61 tiles
constexpr int kBigTilesPerWave = 4;  // 64 tiles over 16 waves: what a wave keeps in registers
constexpr int kBigMaxLayers = 64;
constexpr int kBH = 32;

using bf32x4 = __attribute__((ext_vector_type(4))) float;

struct BigLayer {
    const float* bias;   // of the aggregation this entry stands for, or null
    const float* Wnext;  // [32][64] weights of the NEXT layer's transform, or null after the last hidden aggregation
    int32_t act, pad;
};

struct BigArgs {
    const int32_t* graph_ptr;
    const int32_t* lrow;   // support L: row pointers, diagonal first
    const int32_t* lcol;   // global column ids
    const float* lval;
    const float* Zin;      // [num_nodes][64] Z0 | Z1 of the first aggregation
    float* Hout;           // [num_nodes][32] last hidden activations (row-major)
    uint2* rec;            // [num_graphs][rec_cap]
    int32_t* status;
    int32_t rec_cap, max_nodes, num_hidden;
    int32_t lds_cnt_off, lds_perm_off, lds_stage_off, lds_tab_off;  // byte offsets inside the dynamic LDS; the zero row sits at max_nodes * 128
    BigLayer layers[kBigMaxLayers];
};

__device__ __forceinline__ int big_key(int row) { return (row >> 1) & 3; }
__device__ __forceinline__ unsigned big_word(int u) { return ((unsigned)u << 7) | ((unsigned)big_key(u) << 4); }

typedef __attribute__((address_space(3))) const bf32x4 big_lds_cf4;
__device__ __forceinline__ float4 big_lds_chunk(unsigned addr) {  // Z1 chunk at ABSOLUTE LDS byte address (bufB at LDS offset 0)
    const bf32x4 z = *reinterpret_cast<big_lds_cf4*>(addr);
    return make_float4(z[0], z[1], z[2], z[3]);
}
__device__ __forceinline__ float4 big_fma4(float a, float4 z, float4 acc) {
    acc.x = fmaf(a, z.x, acc.x); acc.y = fmaf(a, z.y, acc.y); acc.z = fmaf(a, z.z, acc.z); acc.w = fmaf(a, z.w, acc.w);
    return acc;
}
__device__ __forceinline__ float big_act(float x, int act) { return apply_act(x, act); }

struct BigRec4 { uint2 r0, r1, r2, r3; };  // the records of four consecutive trips of this lane
__device__ __forceinline__ void big_load_group(BigRec4& G, const char* p) {
    G.r0 = *reinterpret_cast<const uint2*>(p);
    G.r1 = *reinterpret_cast<const uint2*>(p + 512);
    G.r2 = *reinterpret_cast<const uint2*>(p + 1024);
    G.r3 = *reinterpret_cast<const uint2*>(p + 1536);
}

__device__ __forceinline__ void big_load_bfrag(const float* W, float (&b)[8][4]) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) b[s][ct] = W[(4 * s + kq) * 64 + ct * 16 + r];
}

__global__ __launch_bounds__(kBigBlock) __attribute__((amdgpu_waves_per_eu(4))) void k_big(BigArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char big_lds[];
    const int g = blockIdx.x;
    const int n0 = a.graph_ptr[g], ng = a.graph_ptr[g + 1] - n0;
    if (ng <= 0) return;
    float* bufB = reinterpret_cast<float*>(big_lds);  // LDS offset 0: a gather address is the record's word ^ (chunk << 4)
    const unsigned zrow = (unsigned)a.max_nodes * 128u;
    unsigned short* cnt = reinterpret_cast<unsigned short*>(big_lds + a.lds_cnt_off);  // [max_nodes] entries per row (clamped: only orders rows and bounds walks)
    unsigned short* perm = reinterpret_cast<unsigned short*>(big_lds + a.lds_perm_off);  // [max_nodes] rows by descending entry count
    int* hist = reinterpret_cast<int*>(big_lds + a.lds_stage_off);                    // [576] (P0 only: the staging tiles' space)
    int* ttrips = reinterpret_cast<int*>(big_lds + a.lds_tab_off);                    // [64] trips per tile
    unsigned* tbase = reinterpret_cast<unsigned*>(ttrips + 64);                        // [64] first record of a tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles = (ng + 15) >> 4;
    uint2* rec = a.rec + (size_t)g * a.rec_cap;
    int fault = 0;

    // ---- P0: row lengths, row order (counting sort, descending), Z1 of the first aggregation into LDS
    for (int i = threadIdx.x; i < 576; i += kBigBlock) hist[i] = 0;
    if (threadIdx.x < 32) reinterpret_cast<float*>(big_lds + zrow)[threadIdx.x] = 0.f;
    __syncthreads();
    for (int v = threadIdx.x; v < ng; v += kBigBlock) {
        const unsigned c = (unsigned)(a.lrow[n0 + v + 1] - a.lrow[n0 + v]);
        cnt[v] = (unsigned short)min(c, 65535u);
        atomicAdd(&hist[min((int)c, 575)], 1);
    }
    for (int idx = threadIdx.x; idx < ng * 8; idx += kBigBlock) {
        const int v = idx >> 3, c = idx & 7;
        const float4 z = *reinterpret_cast<const float4*>(a.Zin + (size_t)(n0 + v) * 64 + kBH + 4 * c);
        *reinterpret_cast<float4*>(bufB + v * kBH + ((c ^ big_key(v)) << 2)) = z;
    }
    __syncthreads();
    // start offset of count class c in the descending order = rows with a larger count (576 bins: one pass, once per graph)
    int my_off = 0;
    if (threadIdx.x < 576) {
        for (int c = (int)threadIdx.x + 1; c < 576; ++c) my_off += hist[c];
    }
    __syncthreads();
    if (threadIdx.x < 576) hist[threadIdx.x] = my_off;
    __syncthreads();
    for (int v = threadIdx.x; v < ng; v += kBigBlock) {
        const int pos = atomicAdd(&hist[min((int)cnt[v], 575)], 1);
        perm[pos] = (unsigned short)v;  // (order among equal counts: whatever the atomics took - it decides which rows share a pass, never a sum)
    }
    __syncthreads();
    // trips of every tile (its first row is its longest) and where its records start: one wave, one scan
    if (wave == 0) {
        const int tl = lane < tiles ? max(1, (int)((cnt[perm[lane * 16]] + 3) >> 2)) : 0;
        int incl = tl;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        const int total = __shfl(incl, tiles - 1);
        // (a row longer than its graph has vertices - repeated columns in the caller's matrix - could outgrow the slice: no
        // records and a fault bit instead of a write past it)
        const bool fits = (long)total * 64 + 448 <= (long)a.rec_cap;  // (+ the groups read ahead past the last trip)
        ttrips[lane] = fits ? tl : 0;
        tbase[lane] = (unsigned)(incl - tl) * 64u;
        if (!fits && lane == 0) fault |= DGCN_FAULT_DEGREE_RANGE;
    }
    __syncthreads();
    const int s16 = lane >> 2, kq4 = lane & 3;  // aggregation: row slot of the tile, quarter of the row
    // every wave writes the records of its own tiles (read back by the same lanes: no barrier)
    for (int t = wave; t < tiles; t += kBigWaves) {
        const int trips = __builtin_amdgcn_readfirstlane(ttrips[t]);
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)tbase[t]);
        const int slot = t * 16 + s16;
        const bool has = slot < ng;
        const int v = has ? (int)perm[slot] : 0;
        const int start = has ? a.lrow[n0 + v] : 0;
        const int c = has ? a.lrow[n0 + v + 1] - start : 0;
        uint2* out = rec + base + lane;
        for (int tt = 0; tt < trips; ++tt) {
            const int e = 4 * tt + kq4;
            uint2 r = make_uint2(0x80000000u, zrow);  // {-0.0f, zero row}: fmaf(-0.0f, +0.0f, acc) == acc for every acc
            if (e < c) {
                const int u = a.lcol[start + e] - n0;
                if (u < 0 || u >= ng) fault |= DGCN_FAULT_BAD_COLUMN;
                else r = make_uint2(__float_as_uint(a.lval[start + e]), big_word(u));
            }
            out[tt * 64] = r;
        }
    }
    // (the records are read by the lanes that wrote them, after at least one workgroup barrier below)

    // ---- layers
    // H and Z0 never leave the chip: a wave keeps the rows of its (at most four) tiles in registers across the barriers -
    // Z0 in the aggregation's lane layout between a transform and the next aggregation, H' in the MFMA's operand layout
    // between an aggregation and the next transform - and converts between the two layouts through its own 2 KB staging
    // tile in LDS (two ds_write_b128 + eight ds_read_b32 one way, two + two the other; a wave's LDS operations complete in
    // order, so no workgroup barrier is involved).  (First version: both through global scratch - 4 x 128 bytes per vertex
    // and layer, 118 MB per layer for 256 graphs of 900 vertices: the launch ran at the L2 / MALL's pace, 24 us per layer
    // against 3.7 us of LDS time and 6 us of MFMA time.)
    const int cfirst = kq4 | (((s16 >> 1) & 1) << 2), csecond = cfirst ^ 4;  // chunks of this lane (upper half first on odd slot pairs: bank groups)
    const unsigned cA = (unsigned)cfirst << 4, cB = (unsigned)csecond << 4;
    const int mr = lane & 15, mq = lane >> 4;  // transform: row of the tile, k quarter
    float* stg = reinterpret_cast<float*>(big_lds + a.lds_stage_off) + wave * 512;  // [16 rows][32], 16-byte chunks XOR-swizzled by row & 7
    float bfrag[8][4];
    float pz[kBigTilesPerWave][8];
    // Z0 of the first aggregation: from the caller's Z (row-major, Z0 | Z1), in the aggregation's layout
#pragma unroll
    for (int k = 0; k < kBigTilesPerWave; ++k) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pz[k][j] = 0.f;
        const int slot = (wave + kBigWaves * k) * 16 + s16;
        if (slot < ng) {
            const float* zr = a.Zin + (size_t)(n0 + (int)perm[slot]) * 64;
            const float4 yA = *reinterpret_cast<const float4*>(zr + 4 * cfirst), yB = *reinterpret_cast<const float4*>(zr + 4 * csecond);
            pz[k][0] = yA.x; pz[k][1] = yA.y; pz[k][2] = yA.z; pz[k][3] = yA.w;
            pz[k][4] = yB.x; pz[k][5] = yB.y; pz[k][6] = yB.z; pz[k][7] = yB.w;
        }
    }
    const unsigned voff = (unsigned)lane * 8u;
    for (int i = 0; i < a.num_hidden; ++i) {
        const BigLayer& L = a.layers[i];
        const bool last = i == a.num_hidden - 1;
        // -------- aggregation: H' = act(Z0 + L.Z1 + b), 4 lanes x 2 float4 per row, 16 rows per pass.  Every trip needs a
        // record from global memory (L2 / MALL: 500 .. 2 000 cycles), so the records are requested a GROUP of four trips
        // ahead - the next group of this tile, or the first group of the wave's next tile.
        {
            BigRec4 A = {};
            if (wave < tiles) {
                const unsigned base0 = (unsigned)__builtin_amdgcn_readfirstlane((int)tbase[wave]);
                big_load_group(A, reinterpret_cast<const char*>(rec + base0) + voff);
            }
#define DGCN_BQB(x, e) __builtin_amdgcn_update_dpp(0, (int)(x), (e) * 0x55, 0xf, 0xf, true)
#define DGCN_BTRIP(R)                                                                                                   \
            {                                                                                                          \
                float4 zA[4], zB[4];                                                                                   \
                float av[4];                                                                                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                        \
                    const unsigned w = (unsigned)(e == 0 ? DGCN_BQB(R.y, 0) : e == 1 ? DGCN_BQB(R.y, 1) : e == 2 ? DGCN_BQB(R.y, 2) : DGCN_BQB(R.y, 3)); \
                    av[e] = __int_as_float(e == 0 ? DGCN_BQB(R.x, 0) : e == 1 ? DGCN_BQB(R.x, 1) : e == 2 ? DGCN_BQB(R.x, 2) : DGCN_BQB(R.x, 3)); \
                    zA[e] = big_lds_chunk(w ^ cA);                                                                     \
                    zB[e] = big_lds_chunk(w ^ cB);                                                                     \
                }                                                                                                      \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                        \
                    accA = big_fma4(av[e], zA[e], accA);                                                               \
                    accB = big_fma4(av[e], zB[e], accB);                                                               \
                }                                                                                                      \
            }
#pragma unroll
            for (int k = 0; k < kBigTilesPerWave; ++k) {
                const int t = wave + kBigWaves * k;
                if (t < tiles) {  // (wave-uniform)
                    const int tn = t + kBigWaves;
                    const int trips = __builtin_amdgcn_readfirstlane(ttrips[t]);
                    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)tbase[t]);
                    const unsigned base_n = (k + 1 < kBigTilesPerWave && tn < tiles) ? (unsigned)__builtin_amdgcn_readfirstlane((int)tbase[tn]) : 0u;
                    const int slot = t * 16 + s16;
                    const bool has = slot < ng;
                    float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = accA;
                    const char* bp = reinterpret_cast<const char*>(rec + base) + voff;
                    for (int g0 = 0; g0 < trips; g0 += 4) {  // (trips is wave-uniform: scalar branches)
                        BigRec4 Bn = A;
                        if (g0 + 4 < trips) big_load_group(Bn, bp + (size_t)(g0 + 4) * 512);  // (its last trips may lie past the tile: never walked)
                        else if (k + 1 < kBigTilesPerWave && tn < tiles) big_load_group(Bn, reinterpret_cast<const char*>(rec + base_n) + voff);
                        DGCN_BTRIP(A.r0)
                        if (g0 + 1 < trips) DGCN_BTRIP(A.r1)
                        if (g0 + 2 < trips) DGCN_BTRIP(A.r2)
                        if (g0 + 3 < trips) DGCN_BTRIP(A.r3)
                        A = Bn;
                    }
                    float4 oA = make_float4(pz[k][0] + accA.x, pz[k][1] + accA.y, pz[k][2] + accA.z, pz[k][3] + accA.w);
                    float4 oB = make_float4(pz[k][4] + accB.x, pz[k][5] + accB.y, pz[k][6] + accB.z, pz[k][7] + accB.w);
                    if (L.bias) {  // (fetched here, not kept through the walk: eight registers the kernel does not have)
                        const float4 biasA = *reinterpret_cast<const float4*>(L.bias + 4 * cfirst);
                        const float4 biasB = *reinterpret_cast<const float4*>(L.bias + 4 * csecond);
                        oA.x += biasA.x; oA.y += biasA.y; oA.z += biasA.z; oA.w += biasA.w;
                        oB.x += biasB.x; oB.y += biasB.y; oB.z += biasB.z; oB.w += biasB.w;
                    }
                    oA.x = big_act(oA.x, L.act); oA.y = big_act(oA.y, L.act); oA.z = big_act(oA.z, L.act); oA.w = big_act(oA.w, L.act);
                    oB.x = big_act(oB.x, L.act); oB.y = big_act(oB.y, L.act); oB.z = big_act(oB.z, L.act); oB.w = big_act(oB.w, L.act);
                    if (last) {
                        if (has && trips > 0) {
                            float* dst = a.Hout + (size_t)(n0 + (int)perm[slot]) * kBH;
                            *reinterpret_cast<float4*>(dst + 4 * cfirst) = oA;
                            *reinterpret_cast<float4*>(dst + 4 * csecond) = oB;
                        }
                    } else {
                        // aggregation layout -> operand layout of the next transform (lane 16 q + r: H'[r][4 s + q], s = 0..7)
                        *reinterpret_cast<float4*>(stg + s16 * kBH + ((cfirst ^ (s16 & 7)) << 2)) = oA;
                        *reinterpret_cast<float4*>(stg + s16 * kBH + ((csecond ^ (s16 & 7)) << 2)) = oB;
                        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int s = 0; s < 8; ++s) pz[k][s] = stg[mr * kBH + (((s ^ (mr & 7)) << 2) | mq)];
                        __builtin_amdgcn_s_waitcnt(0xC07F);
                        __builtin_amdgcn_wave_barrier();  // (the staging tile is rewritten by the next tile)
                    }
                }
            }
#undef DGCN_BTRIP
#undef DGCN_BQB
        }
        if (last) break;
        // the next layer's weight fragments: requested here, they land while this wave waits at the barrier
        big_load_bfrag(L.Wnext, bfrag);
        __syncthreads();  // every gather of this layer has read Z1
        // -------- transform of the next layer: Z0 | Z1 = H'.[W0 | W1], v_mfma_f32_16x16x4_f32, operands swapped (D^T = W^T.H^T)
        // so that a lane ends with four consecutive features of one vertex; Z1 -> bufB, Z0 -> registers (aggregation layout)
#pragma unroll
        for (int k = 0; k < kBigTilesPerWave; ++k) {
            const int t = wave + kBigWaves * k;
            if (t < tiles) {
                bf32x4 acc[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct] = (bf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfrag[s][ct], pz[k][s], acc[ct], 0, 0, 0);
                const int slot = t * 16 + mr;
                if (slot < ng) {
                    const int v = (int)perm[slot];
#pragma unroll
                    for (int ct = 2; ct < 4; ++ct) {
                        const int chunk = (ct & 1) * 4 + mq;
                        *reinterpret_cast<float4*>(bufB + v * kBH + ((chunk ^ big_key(v)) << 2)) = make_float4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]);
                    }
                }
                // Z0: MFMA layout (row mr, chunks mq and 4 + mq) -> aggregation layout (row s16, chunks cfirst / csecond)
                *reinterpret_cast<float4*>(stg + mr * kBH + ((mq ^ (mr & 7)) << 2)) = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]);
                *reinterpret_cast<float4*>(stg + mr * kBH + (((4 + mq) ^ (mr & 7)) << 2)) = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
                const float4 yA = *reinterpret_cast<const float4*>(stg + s16 * kBH + ((cfirst ^ (s16 & 7)) << 2));
                const float4 yB = *reinterpret_cast<const float4*>(stg + s16 * kBH + ((csecond ^ (s16 & 7)) << 2));
                pz[k][0] = yA.x; pz[k][1] = yA.y; pz[k][2] = yA.z; pz[k][3] = yA.w;
                pz[k][4] = yB.x; pz[k][5] = yB.y; pz[k][6] = yB.z; pz[k][7] = yB.w;
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();  // Z1 of the next layer is complete
    }
    if (fault) atomicOr(a.status, fault);
}

// ---------------------------------------------------------------------------------------------------------------------
int spmm_dispatch(const DgcnCsr* S, const int32_t* graph_ptr, int B, int max_nodes, const float* Z, int ldz, int C,
                  const float* Y0, int ldy0, const float* bias, int act, float* Y, int ldy, hipStream_t s);
int transform_dispatch(const float* H, int ldh, float h_const, int rows, int cin, const float* W, int ctot, float* Z,
                       int ldz, hipStream_t s);
int spmm_f64acc_dispatch(const DgcnCsr* S, const int32_t* graph_ptr, int B, int max_nodes, const float* Z, int ldz, int C,
                         const float* Y0, int ldy0, const float* bias, int act, float* Y, int ldy, hipStream_t s);
int transform_f64acc_dispatch(const float* H, int ldh, float h_const, int rows, int cin, const float* W, int ctot, float* Z,
                              int ldz, hipStream_t s);

static size_t b256(size_t x) { return (x + 255) & ~(size_t)255; }

static int big_rec_cap(const DgcnBatch* b) {
    // fused.hip's bound for block-major records of rows in descending order: entries + 20 N + 192 (+ one cache line)
    return ((b->max_graph_edges + b->max_nodes + 2 + 16 + 15) & ~15) + ((20 * b->max_nodes + 448 + 15) & ~15);
}

static size_t big_lds_bytes(int max_nodes, int* cnt_off, int* perm_off, int* stage_off, int* tab_off) {
    size_t off = (size_t)max_nodes * 128 + 128;  // Z1 + the zero row
    *stage_off = (int)off;
    off += (size_t)kBigWaves * 2048;            // a 16 x 32 float tile per wave (P0: the count histogram)
    *cnt_off = (int)off;
    off += ((size_t)max_nodes * 2 + 15) & ~(size_t)15;
    *perm_off = (int)off;
    off += ((size_t)max_nodes * 2 + 15) & ~(size_t)15;
    *tab_off = (int)off;
    off += 128 * 4;
    return off;
}

// 1 = a deep [I, L] stack F -> 32 -> .. -> 32 -> 1 on graphs of at most 976 vertices: the shape k_big takes
int big_takes(const DgcnBatch* b, const DgcnModel* m) {
    if (const char* e = getenv("DGCN_BIG")) if (atoi(e) == 0) return 0;
    if (!b || !m || !m->layers_host || m->num_supports != 2 || m->num_layers < 3 || m->num_layers - 2 > kBigMaxLayers) return 0;
    if (b->max_nodes <= 0 || b->max_nodes > kBigMaxNodes) return 0;
    const int Lc = m->num_layers;
    for (int l = 0; l < Lc; ++l) {
        const DgcnLayer& L = m->layers_host[l];
        if (!L.weights || L.in_dim <= 0) return 0;
        if (l > 0 && L.in_dim != kBH) return 0;
        if (l < Lc - 1 && L.out_dim != kBH) return 0;
        if (l == Lc - 1 && L.out_dim != 1) return 0;
        if (L.bias && ((uintptr_t)L.bias & 15)) return 0;
    }
    return 1;
}

// scratch beyond the layer-by-layer buffers (Z twice, H): the records
size_t big_workspace(const DgcnBatch* b, const DgcnModel* m) {
    if (!big_takes(b, m)) return 0;
    const size_t B = (size_t)std::max(b->num_graphs, 1);
    return 256 + b256(B * (size_t)big_rec_cap(b) * 8);
}

// The forward pass with the hidden stack in one launch: layer 0 and the transform of layer 1 (chains in double) and the
// last layer (32 -> 1) by the layer-by-layer kernels, exactly as layered_forward runs them; layers 1 .. L-2 by k_big.
// `lws`: dgcn_gcn_forward_workspace(b, m, 0) bytes (Z twice, H), `bws`: big_workspace(b, m) bytes.
int big_forward(const DgcnBatch* b, const DgcnCsr* lap, const DgcnModel* m, const float* X, float x_const, float* scores,
                void* lws, void* bws, int32_t* status, hipStream_t s) {
    const size_t n = (size_t)b->num_nodes;
    const int Lc = m->num_layers;
    char* w0 = reinterpret_cast<char*>(lws);
    const size_t zsz = b256(n * 2 * kBH * sizeof(float));
    float* Zbuf = reinterpret_cast<float*>(w0);
    float* Hbuf = reinterpret_cast<float*>(w0 + 2 * zsz);
    char* w1 = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(bws) + 255) & ~(uintptr_t)255);
    uint2* rec = reinterpret_cast<uint2*>(w1);
    const DgcnLayer& L0 = m->layers_host[0];
    const DgcnLayer& L1 = m->layers_host[1];
    int rc = transform_dispatch(X, L0.in_dim, x_const, b->num_nodes, L0.in_dim, L0.weights, 2 * kBH, Zbuf, 2 * kBH, s);
    if (rc) return rc;
    rc = spmm_f64acc_dispatch(lap, b->graph_ptr, b->num_graphs, b->max_nodes, Zbuf + kBH, 2 * kBH, kBH, Zbuf, 2 * kBH, L0.bias, L0.act,
                              Hbuf, kBH, s);
    if (rc) return rc;
    rc = transform_f64acc_dispatch(Hbuf, kBH, x_const, b->num_nodes, kBH, L1.weights, 2 * kBH, Zbuf, 2 * kBH, s);
    if (rc) return rc;
    BigArgs a = {};
    a.graph_ptr = b->graph_ptr;
    a.lrow = lap->row_ptr; a.lcol = lap->col_idx; a.lval = lap->values;
    a.Zin = Zbuf; a.Hout = Hbuf; a.rec = rec; a.status = status;
    a.rec_cap = big_rec_cap(b);
    a.max_nodes = (std::max(b->max_nodes, 16) + 15) & ~15;
    a.num_hidden = Lc - 2;
    for (int l = 1; l <= Lc - 2; ++l) {
        const DgcnLayer& L = m->layers_host[l];
        a.layers[l - 1].bias = L.bias;
        a.layers[l - 1].act = L.act;
        a.layers[l - 1].Wnext = l < Lc - 2 ? m->layers_host[l + 1].weights : nullptr;
    }
    const size_t lds = big_lds_bytes(a.max_nodes, &a.lds_cnt_off, &a.lds_perm_off, &a.lds_stage_off, &a.lds_tab_off);
    if (lds > 160 * 1024) return fail(DGCN_ERR_UNSUPPORTED, "k_big: %zu bytes of LDS", lds);
    if (lds > 64 * 1024) {
        static std::atomic<int> reserved[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!reserved[dev & 63].load(std::memory_order_relaxed)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_big), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return fail(DGCN_ERR_LAUNCH, "k_big: cannot reserve %zu bytes of LDS", lds);
            reserved[dev & 63].store(1, std::memory_order_relaxed);
        }
    }
    {
        TimedLaunch t("big_stack", s);
        DGCN_LAUNCH(t, k_big, dim3((unsigned)b->num_graphs), dim3(kBigBlock), lds, s, a);
        if ((rc = check_launch("k_big"))) return rc;
    }
    const DgcnLayer& LL = m->layers_host[Lc - 1];
    rc = transform_dispatch(Hbuf, kBH, x_const, b->num_nodes, kBH, LL.weights, 2, Zbuf, 2, s);
    if (rc) return rc;
    return spmm_dispatch(lap, b->graph_ptr, b->num_graphs, b->max_nodes, Zbuf + 1, 2, 1, Zbuf, 2, LL.bias, LL.act, scores, 1, s);
}

}  // namespace dgcn
