// k_big2: the deep c32 stack in ONE launch for graphs of 977 .. 1 920 vertices - the joint conflict graph of K = 3 channels x 500
// flows (wireless_dqn_test_mc.py:37, 161, 244-289; wireless_rollout_test_flood.py:98-133) and whatever else outgrows k_big.
//
// k_big (big.hip) keeps Z1 - what an aggregation gathers at random, 128 bytes per vertex - whole in LDS: 976 vertices.  Beyond
// that these graphs ran layer by layer (3 - 4 x the time per vertex, 128 x N = 1 500: 1.30 ms per step).  Here Z1 lives in LDS a
// FEATURE HALF at a time (64 bytes per vertex: 1 920 vertices next to the staging tiles) and every aggregation walks its
// records twice, once per half; a chain is per (row, feature), so its order - the row's entries in storage order from 0 - is
// untouched and the bits equal every other path's (include/dgcn.h "Precision"; tests/test_gpu_big2.py).
//
//   LDS        Z1h[N][16] (the half in use), a zero row, a 2 KB staging tile per wave, the row order
//   registers  of the (at most fifteen) 16-row tiles a wave owns: H (32 features, the MFMA's operand layout) from one
//              aggregation's epilogue to the next one's, and the lo half of the running sum between the two walks.  ONE
//              512-thread workgroup per CU: eight waves with 256 registers each - what holds 12 registers per tile.
//   global     the support as block-major 8-byte records {value, LDS address of the neighbour's Z1h row} (big.hip's)
//
// The transform Z = H.[W0 | W1] is cut in three, each computed from H when its output is needed (v_mfma_f32_16x16x4_f32, one
// sixteen-column tile at a time; layer index 1: v_mfma_f64_16x16x4_f64, the contract's double chains):
//   S1  Z1 lo = H.W1[:, 0:16]  -> LDS     | barrier |  S2  walk: lo sums            | barrier |
//   S3  Z1 hi = H.W1[:, 16:32] -> LDS     | barrier |  S4  walk: hi sums; Z0 = H.W0; H' = act(Z0 + sums + b) -> operand layout
// Four barriers per layer instead of two; S4 mixes gathers and MFMAs of different waves, so the LDS array and the matrix
// pipes overlap there.  The first layer on constant features, the last layer (32 -> 1) and the local greedy search run inside
// the launch as in k_big (the search as lgs_rounds.h's rounds: a mask per vertex would not fit).
#include <algorithm>
#include <atomic>

#include "common.h"
#include "lgs_rounds.h"
#include "big_common.h"
#include "cand_select.h"
#include "rollout_bits.h"

namespace dgcn {

constexpr int kB2Block = 512;
constexpr int kB2Waves = kB2Block / 64;
constexpr int kB2MaxNodes = 1920;  // 120 tiles: fifteen per wave; Z1h 120 KB + 16 KB of staging + the row tables
constexpr int kB2Bins = 2048;      // one bin per entry count of a row (a row of a 1 920-vertex graph has at most 1 921 entries)
constexpr int kB2MaxTiles = 128;

__device__ __forceinline__ unsigned b2_word(int u) { return (unsigned)u << 6; }

// one sixteen-column tile of the weights as the MFMA's B fragments (big_load_bfrag's mapping, one column tile)
__device__ __forceinline__ void b2_load_bcol(const float* W, int ct, bool f64map, float (&b)[8]) {
    const int lane = threadIdx.x & 63;
    const int r0 = lane & 15, kq = lane >> 4;
    const int r = f64map ? 4 * (r0 & 3) + (r0 >> 2) : r0;
#pragma unroll
    for (int s = 0; s < 8; ++s) b[s] = W[(4 * s + kq) * 64 + ct * 16 + r];
}

// sixteen rows x sixteen output columns: a lane ends with four consecutive features (4 mq ..) of row mr
__device__ __forceinline__ float4 b2_mma(const float (&b)[8], const float (&h)[8], bool f64) {
    if (f64) {
        bf64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b[s], (double)h[s], acc, 0, 0, 0);
        return make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
    }
    bf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], h[s], acc, 0, 0, 0);
    return make_float4(acc[0], acc[1], acc[2], acc[3]);
}

#ifdef DGCN_DIAG
#define B2_STAMP(i)                                                    \
    do {                                                               \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();    \
        b2_acc[i] += _t - b2_t0;                                       \
        b2_t0 = _t;                                                    \
    } while (0)
#else
#define B2_STAMP(i) do { } while (0)
#endif

// the same for two / three independent products at once: a single chain of eight dependent MFMAs leaves the matrix pipe idle
// between issues (two waves per SIMD), interleaved chains fill it
__device__ __forceinline__ void b2_mma2(const float (&b)[8], const float (&h0)[8], const float (&h1)[8], bool f64, float4& o0, float4& o1) {
    if (f64) {
        bf64x4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b[s], (double)h0[s], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b[s], (double)h1[s], a1, 0, 0, 0);
        }
        o0 = make_float4((float)a0[0], (float)a0[1], (float)a0[2], (float)a0[3]);
        o1 = make_float4((float)a1[0], (float)a1[1], (float)a1[2], (float)a1[3]);
        return;
    }
    bf32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], h0[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s], h1[s], a1, 0, 0, 0);
    }
    o0 = make_float4(a0[0], a0[1], a0[2], a0[3]);
    o1 = make_float4(a1[0], a1[1], a1[2], a1[3]);
}
__device__ __forceinline__ void b2_mma2b(const float (&b0)[8], const float (&b1)[8], const float (&h)[8], bool f64, float4& o0, float4& o1) {
    if (f64) {
        bf64x4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const double hd = (double)h[s];
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b0[s], hd, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b1[s], hd, a1, 0, 0, 0);
        }
        o0 = make_float4((float)a0[0], (float)a0[1], (float)a0[2], (float)a0[3]);
        o1 = make_float4((float)a1[0], (float)a1[1], (float)a1[2], (float)a1[3]);
        return;
    }
    bf32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[s], h[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[s], h[s], a1, 0, 0, 0);
    }
    o0 = make_float4(a0[0], a0[1], a0[2], a0[3]);
    o1 = make_float4(a1[0], a1[1], a1[2], a1[3]);
}
__device__ __forceinline__ void b2_mma3(const float (&b0)[8], const float (&b1)[8], const float (&b2)[8], const float (&h)[8], bool f64,
                                        float4& o0, float4& o1, float4& o2) {
    if (f64) {
        bf64x4 a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0}, a2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const double hd = (double)h[s];
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b0[s], hd, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b1[s], hd, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b2[s], hd, a2, 0, 0, 0);
        }
        o0 = make_float4((float)a0[0], (float)a0[1], (float)a0[2], (float)a0[3]);
        o1 = make_float4((float)a1[0], (float)a1[1], (float)a1[2], (float)a1[3]);
        o2 = make_float4((float)a2[0], (float)a2[1], (float)a2[2], (float)a2[3]);
        return;
    }
    bf32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[s], h[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[s], h[s], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b2[s], h[s], a2, 0, 0, 0);
    }
    o0 = make_float4(a0[0], a0[1], a0[2], a0[3]);
    o1 = make_float4(a1[0], a1[1], a1[2], a1[3]);
    o2 = make_float4(a2[0], a2[1], a2[2], a2[3]);
}

// RESID: compiled with the residual-step code (big2_residual; big.hip's k_big<.., RESID> is the model): `state` is the running
// state, the residual graph's support is formed from the adjacency and the state while the records are written, the greedy step
// (rounds / central pick / priorities out) runs on the undecided vertices
template <int TILES, bool RESID = false>
__global__ __launch_bounds__(kB2Block) void k_big2(BigArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char b2_lds[];
    const int g = blockIdx.x;
    const int n0 = a.graph_ptr[g], ng = a.graph_ptr[g + 1] - n0;
    if (RESID && a.cid && threadIdx.x < kCandMaxBeam) a.cid[(size_t)g * kCandMaxBeam + threadIdx.x] = -1;  // (a graph left alone has no candidates)
    if (ng <= 0) {
        if (a.do_lgs && threadIdx.x == 0) {
            if (a.rounds) a.rounds[g] = 0;
            if (a.totals) a.totals[g] = 0.0;
        }
        return;
    }
    float* bufH = reinterpret_cast<float*>(b2_lds);  // LDS offset 0: a gather address is the record's word | (chunk << 4)
    const unsigned zrow = (unsigned)a.max_nodes * 64u;
    unsigned short* cnt = reinterpret_cast<unsigned short*>(b2_lds + a.lds_cnt_off);
    unsigned short* perm = reinterpret_cast<unsigned short*>(b2_lds + a.lds_perm_off);
    int* hist = reinterpret_cast<int*>(b2_lds + a.lds_stage_off);  // [kB2Bins] (P0 only: the staging tiles' space)
    int* ttrips = reinterpret_cast<int*>(b2_lds + a.lds_tab_off);  // [kB2MaxTiles] trips per tile
    unsigned* tbase = reinterpret_cast<unsigned*>(ttrips + kB2MaxTiles);  // [kB2MaxTiles] first record of a tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2* rec = a.rec + (size_t)g * a.rec_cap;
    int fault = 0;
    constexpr bool resid = RESID;
    uint8_t* alive = reinterpret_cast<uint8_t*>(b2_lds);  // (residual steps) in the Z1h space until S1 of the first layer writes there
    int nr = ng;  // rows of the (residual) graph: its undecided vertices
    if (resid) {
        if (threadIdx.x < 2) ttrips[threadIdx.x] = 0;
        __syncthreads();
        int c_al = 0, pos = 0;
        for (int v = threadIdx.x; v < ng; v += kB2Block) {
            const bool al = a.state[n0 + v] == 0;
            alive[v] = al ? 1 : 0;
            c_al += al;
            pos |= al && (a.weights ? a.weights[n0 + v] : 1.0) > 0.0;
        }
        if (c_al) atomicAdd(&ttrips[0], c_al);
        if (pos) atomicOr(&ttrips[1], 1);
        __syncthreads();
        nr = ttrips[0];
        const int any_pos = ttrips[1];
        __syncthreads();
        // nothing left, or no positive weight left (np.sum(wts_nn) <= 0 -> break, mwis_gdpg_call.py:286): the graph is left alone
        if (!any_pos) {
            if (threadIdx.x == 0) {
                if (a.rounds) a.rounds[g] = 0;
                if (a.totals) a.totals[g] = 0.0;
                if (a.active) a.active[g] = 0;
            }
            for (int v = threadIdx.x; v < ng; v += kB2Block) a.scores[n0 + v] = 0.f;
            return;
        }
    }
    const int tiles = (nr + 15) >> 4;
#ifdef DGCN_DIAG
    unsigned long long b2_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long b2_acc[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)g * 16 + 12] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- P0: row lengths, row order (counting sort, descending; big.hip's, one bin per count)
    for (int i = threadIdx.x; i < kB2Bins; i += kB2Block) hist[i] = 0;
    if (threadIdx.x < 16) reinterpret_cast<float*>(b2_lds + zrow)[threadIdx.x] = 0.f;
    __syncthreads();
    for (int v = threadIdx.x; v < ng; v += kB2Block) {
        unsigned c = a.arow ? (unsigned)(a.arow[n0 + v + 1] - a.arow[n0 + v]) + 1u : (unsigned)(a.lrow[n0 + v + 1] - a.lrow[n0 + v]);
        if (resid) {  // degree in the residual graph; a decided vertex has no row (count 0: it sorts behind every undecided one)
            c = 0;
            if (alive[v]) {
                c = 1;
                const int rs = a.arow[n0 + v], re = a.arow[n0 + v + 1];
#pragma unroll 4
                for (int j = rs; j < re; ++j) {
                    const int u = a.acol[j] - n0;
                    if ((unsigned)u < (unsigned)ng) c += alive[u];
                }
            }
        }
        cnt[v] = (unsigned short)min(c, 65535u);
        atomicAdd(&hist[min((int)c, kB2Bins - 1)], 1);
    }
    __syncthreads();
    {
        constexpr int PER = kB2Bins / kB2Block;
        int h[PER], own = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { h[j] = hist[threadIdx.x * PER + j]; own += h[j]; }
        int suf = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_down(suf, off);
            if (lane + off < 64) suf += t;
        }
        if (lane == 0) ttrips[wave] = suf;
        __syncthreads();
        int above = suf - own;
        for (int w = wave + 1; w < kB2Waves; ++w) above += ttrips[w];
        __syncthreads();
#pragma unroll
        for (int j = PER - 1; j >= 0; --j) { hist[threadIdx.x * PER + j] = above; above += h[j]; }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < ng; v += kB2Block) {
        const int pos = atomicAdd(&hist[min((int)cnt[v], kB2Bins - 1)], 1);
        perm[pos] = (unsigned short)v;
    }
    __syncthreads();
    // trips of every tile (its longest row's) and where its records start: one wave, two tiles per lane
    if (wave == 0) {
        int tl[2], incl[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int tt = lane + 64 * r;
            int longest = 0;
            if (tt < tiles) {
#pragma unroll 4
                for (int j = 0; j < 16; ++j) {
                    const int sl = tt * 16 + j;
                    if (sl < nr) longest = max(longest, (int)cnt[perm[sl]]);
                }
            }
            tl[r] = tt < tiles ? max(1, (longest + 3) >> 2) : 0;
            incl[r] = tl[r];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl[r], off);
                if (lane >= off) incl[r] += t;
            }
        }
        incl[1] += __shfl(incl[0], 63);
        const int total = __shfl(incl[1], 63);
        const bool fits = (long)total * 64 + 448 <= (long)a.rec_cap;  // (+ the groups read ahead past the last trip)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            ttrips[lane + 64 * r] = fits ? tl[r] : 0;
            tbase[lane + 64 * r] = (unsigned)(incl[r] - tl[r]) * 64u;
        }
        if (!fits && lane == 0) fault |= DGCN_FAULT_DEGREE_RANGE;
    }
    double* dvl = reinterpret_cast<double*>(b2_lds + a.lds_stage_off);  // d^-1/2 per vertex (adjacency input; until the first staging)
    if (a.arow) {
        for (int v = threadIdx.x; v < ng; v += kB2Block) {
            const int d = (int)cnt[v] - 1;
            double x = 0.0;
            if (d >= a.table_len) fault |= DGCN_FAULT_DEGREE_RANGE;
            else if (d >= 0) x = a.dinv[d];  // (a decided vertex of a residual step has no row: nobody reads its slot)
            dvl[v] = x;
        }
    }
    __syncthreads();
    B2_STAMP(0);  // row lengths, row order, tiles, d^-1/2
    const int s16 = lane >> 2, kq4 = lane & 3;  // aggregation: row slot of the tile, chunk of the half
    // ---- the records of this wave's tiles (big.hip's writer; the word is the neighbour's row in the 64-byte-per-vertex half)
    for (int t = wave; t < tiles; t += kB2Waves) {
        const int trips = __builtin_amdgcn_readfirstlane(ttrips[t]);
        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)tbase[t]);
        const int slot = t * 16 + s16;
        const bool has = slot < nr;
        const int v = has ? (int)perm[slot] : 0;
        const int start = has ? (a.arow ? a.arow[n0 + v] : a.lrow[n0 + v]) : 0;
        const int c = has ? (a.arow ? a.arow[n0 + v + 1] - start + 1 : a.lrow[n0 + v + 1] - start) : 0;
        const double dv = (a.arow && has) ? dvl[v] : 0.0;
        uint2* out = rec + base + lane;
        const uint2 nothing = make_uint2(0x80000000u, zrow);  // {-0.0f, zero row}: fmaf(-0.0f, +0.0f, acc) == acc for every acc
        if (resid) {
            // the row of the RESIDUAL graph: the diagonal, then the undecided neighbours in CSR order, compacted by a prefix count
            // within the row's quad (big.hip: k_big<.., RESID>)
            uint2* rowout = rec + base + s16 * 4;
            const int raw = has ? c - 1 : 0;
            if (has && kq4 == 0) rowout[0] = make_uint2(__float_as_uint(1.0f), b2_word(v));
            int pos = has ? 1 : 0;
            for (int j0 = 0; __any(j0 < raw); j0 += 4) {
                const int j = j0 + kq4;
                int u = -1;
                if (j < raw) {
                    u = a.acol[start + j] - n0;
                    if ((unsigned)u >= (unsigned)ng) { fault |= DGCN_FAULT_BAD_COLUMN; u = -1; }
                    else {
                        if (u == v) fault |= DGCN_FAULT_SELF_LOOP;
                        if (!alive[u]) u = -1;
                    }
                }
                const bool keep = u >= 0;
                const unsigned q = (unsigned)(__ballot(keep) >> (lane & ~3)) & 0xfu;
                if (keep) {
                    const int p = pos + __popc(q & ((1u << kq4) - 1u));
                    if (p < 4 * trips) rowout[(p >> 2) * 64 + (p & 3)] = make_uint2(__float_as_uint((float)(-(dvl[u] * dv))), b2_word(u));
                }
                pos += __popc(q);
            }
            for (int p = pos + kq4; p < 4 * trips; p += 4) rowout[(p >> 2) * 64 + (p & 3)] = nothing;
        } else
        for (int t0 = 0; t0 < trips; t0 += 4) {
            int uu[4];
            float vv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * (t0 + i) + kq4;
                uu[i] = -1;
                vv[i] = 0.f;
                if (t0 + i < trips && e < c) {
                    if (a.arow) { if (e > 0) uu[i] = a.acol[start + e - 1] - n0; }
                    else { uu[i] = a.lcol[start + e] - n0; vv[i] = a.lval[start + e]; }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * (t0 + i) + kq4;
                if (t0 + i >= trips) break;
                uint2 r = nothing;
                if (e < c) {
                    if (a.arow && e == 0) r = make_uint2(__float_as_uint(1.0f), b2_word(v));
                    else {
                        const int u = uu[i];
                        if (u < 0 || u >= ng) fault |= DGCN_FAULT_BAD_COLUMN;
                        else if (a.arow) {
                            if (u == v) fault |= DGCN_FAULT_SELF_LOOP;
                            r = make_uint2(__float_as_uint((float)(-(dvl[u] * dv))), b2_word(u));
                        } else r = make_uint2(__float_as_uint(vv[i]), b2_word(u));
                    }
                }
                out[(t0 + i) * 64] = r;
            }
        }
    }
    __syncthreads();  // (the d^-1/2 array shares the staging tiles' space)
    B2_STAMP(1);  // records

    // ---- layers
    const int cfirst = kq4, csecond = kq4 + 4;  // this lane's chunks of its row: one of each half
    const unsigned cOff = (unsigned)kq4 << 4;
    const int mr = lane & 15, mq = lane >> 4;   // transform: row of the tile, k quarter / output chunk
    float* stg = reinterpret_cast<float*>(b2_lds + a.lds_stage_off) + wave * 512;  // [16 rows][32], 16-byte chunks XOR-swizzled by row & 7
    using vtile = float __attribute__((ext_vector_type(TILES == 15 ? 16 : TILES)));  // (a 15-wide vector is not a register tuple)
    vtile pzv[8];   // H of this wave's tiles (operand layout: lane 16 q + r holds H[r][4 s + q]): pzv[s][k], s = 0..7, tile k
    vtile alo[4];   // the lo half of the running row sums: alo[j][k] (STASH: in global scratch instead)
    // Beyond twelve tiles per wave H alone takes 120 of the 256 registers: the lo sums of a tile (a float4 per lane) wait for S3
    // in a slice of the caller's scratch instead - written and read back by the same lane, 64 bytes per vertex and layer through
    // the L2.  (With them in registers the fifteen-tile kernel spilled 276 registers: ER(1 900, 0.004) 5.3 ms per 256 graphs.)
    constexpr bool STASH = TILES > 12;
    float4* stash = a.stash + (size_t)g * ((size_t)a.max_nodes * 4) + lane;  // [tile][64 lanes]
#define B2_LO_PUT(K, V) { if constexpr (STASH) stash[(size_t)(wave + kB2Waves * (K)) * 64] = V; else B2_SET4(alo, K, V) }
#define B2_LO_GET(K, V) { if constexpr (STASH) V = stash[(size_t)(wave + kB2Waves * (K)) * 64]; else B2_GET4(alo, K, V) }
    const unsigned voff = (unsigned)lane * 8u;
    // groups of records in flight ahead of the walk: two where the registers allow (eight tiles per wave), one beyond
    constexpr int DEPTH = TILES <= 8 ? 2 : 1;
    BigRec4 A = {}, Bq = {}, Cq = {};
    int pf_k = 0, pf_j = 0;
    (void)Cq;
#define B2_QB(x, e) __builtin_amdgcn_update_dpp(0, (int)(x), (e) * 0x55, 0xf, 0xf, true)
#define B2_QBF(x, e) __int_as_float(B2_QB(__float_as_int(x), e))
    // (every tile iteration re-derives its LDS addresses from `lane` laundered through an empty asm: left to itself the compiler
    // hoists some thirty of them out of the layer loop, where they cost the registers H needs)
#define B2_LAUNDER                                                                                                       \
        int lane_l = lane;                                                                                             \
        asm volatile("" : "+v"(lane_l));                                                                               \
        const int s16 = lane_l >> 2, kq4 = lane_l & 3, mr = lane_l & 15, mq = lane_l >> 4;                             \
        const int cfirst = kq4, csecond = kq4 + 4;                                                                     \
        const unsigned cOff = (unsigned)kq4 << 4;                                                                      \
        (void)s16; (void)kq4; (void)mr; (void)mq; (void)cfirst; (void)csecond; (void)cOff;
#define B2_TILE_HEAD                                                                                                     \
        B2_LAUNDER                                                                                                     \
        const int t = wave + kB2Waves * k;                                                                             \
        const int trips = __builtin_amdgcn_readfirstlane(ttrips[t]);                                                   \
        const int slot = t * 16 + s16;                                                                                 \
        const bool has = slot < nr;                                                                                    \
        (void)has;
    // the records of a wave's tiles as one stream of groups (four trips each), requested two groups ahead of the walk (big.hip)
#define B2_PREFETCH(X)                                                                                                   \
        {                                                                                                              \
            const int pt_ = wave + kB2Waves * pf_k;                                                                    \
            if (pf_k < TILES && pt_ < tiles) {                                                                         \
                const int ptr_ = __builtin_amdgcn_readfirstlane(ttrips[pt_]);                                          \
                const unsigned pb_ = (unsigned)__builtin_amdgcn_readfirstlane((int)tbase[pt_]);                        \
                big_load_group(X, reinterpret_cast<const char*>(rec + pb_) + (size_t)pf_j * 2048 + voff);              \
                pf_j += 1;                                                                                             \
                if (pf_j * 4 >= ptr_) { pf_j = 0; pf_k += 1; }                                                         \
            }                                                                                                          \
        }
#define B2_WALK(TRIP)                                                                                                    \
        for (int g0 = 0; g0 < trips; g0 += 4) {                                                                        \
            if constexpr (DEPTH == 2) { B2_PREFETCH(Cq) } else { B2_PREFETCH(Bq) }                                     \
            TRIP(A.r0, g0)                                                                                             \
            if (g0 + 1 < trips) TRIP(A.r1, g0 + 1)                                                                     \
            if (g0 + 2 < trips) TRIP(A.r2, g0 + 2)                                                                     \
            if (g0 + 3 < trips) TRIP(A.r3, g0 + 3)                                                                     \
            A = Bq;                                                                                                    \
            if constexpr (DEPTH == 2) Bq = Cq;                                                                         \
        }
#define B2_FIRST_GROUP                                                                                                   \
        pf_k = 0;                                                                                                      \
        pf_j = 0;                                                                                                      \
        B2_PREFETCH(A)                                                                                                 \
        if constexpr (DEPTH == 2) { B2_PREFETCH(Bq) }
    // The tile loops below are NOT unrolled (fifteen copies of every walk would be a megabyte of code).  A tile's registers are
    // picked with a wave-uniform DYNAMIC index into register-resident vectors - pzv[s] holds H[..][4 s + q] of all TILES tiles,
    // one lane of the vector per tile - which the compiler turns into relative register moves (s_set_gpr_idx / v_movrel): one
    // move per value.  (First forms: a chain of compares with constant indices - as selects every tile's registers were live in
    // every iteration and the spilled ones cost a scratch round trip per tile, as branches two dozen taken branches per pick:
    // 0.4 us per tile and phase.)
#define B2_GET_H(K, H) _Pragma("unroll") for (int s = 0; s < 8; ++s) H[s] = pzv[s][K];
#define B2_SET_H(K, H) _Pragma("unroll") for (int s = 0; s < 8; ++s) pzv[s][K] = H[s];
#define B2_GET4(ARR, K, V) { V.x = ARR[0][K]; V.y = ARR[1][K]; V.z = ARR[2][K]; V.w = ARR[3][K]; }
#define B2_SET4(ARR, K, V) { ARR[0][K] = V.x; ARR[1][K] = V.y; ARR[2][K] = V.z; ARR[3][K] = V.w; }
    // aggregation layout (row s16, chunks cfirst / csecond: OA, OB) -> the MFMA's operand layout, into the tile's lane of pzv (through hh)
#define B2_STAGE_TO_OPERAND(OA, OB)                                                                                      \
        *reinterpret_cast<float4*>(stg + s16 * kBH + ((cfirst ^ (s16 & 7)) << 2)) = OA;                                \
        *reinterpret_cast<float4*>(stg + s16 * kBH + ((csecond ^ (s16 & 7)) << 2)) = OB;                               \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */                                                           \
        __builtin_amdgcn_wave_barrier();                                                                               \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) hh[s] = stg[mr * kBH + (((s ^ (mr & 7)) << 2) | mq)];            \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        __builtin_amdgcn_wave_barrier();                                                                               \
        B2_SET_H(k, hh)
    // one half of an aggregation: four entries of the row per trip, one 16-byte chunk of each neighbour's Z1h row
#define B2_TRIP(R, TT)                                                                                                   \
        {                                                                                                              \
            float4 zq[4];                                                                                              \
            float av[4];                                                                                               \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                            \
                const unsigned w = (unsigned)(e == 0 ? B2_QB(R.y, 0) : e == 1 ? B2_QB(R.y, 1) : e == 2 ? B2_QB(R.y, 2) : B2_QB(R.y, 3)); \
                av[e] = __int_as_float(e == 0 ? B2_QB(R.x, 0) : e == 1 ? B2_QB(R.x, 1) : e == 2 ? B2_QB(R.x, 2) : B2_QB(R.x, 3)); \
                zq[e] = big_lds_chunk(w | cOff);                                                                       \
            }                                                                                                          \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) acc = big_fma4(av[e], zq[e], acc);                           \
        }
#pragma unroll
    for (int j = 0; j < 8; ++j) pzv[j] = (vtile)(0.f);

    if (a.front) {
        // -------- layer index 0 on constant input features: the entries' VALUES only, chains in double (big.hip's front)
        const BigFront& F = a.first;
        B2_FIRST_GROUP
#pragma unroll 1
        for (int k = 0; k < TILES; ++k) {
            if (wave + kB2Waves * k < tiles) {  // (wave-uniform)
                B2_TILE_HEAD
                float hh[8];
                // (every row of Z = x.[W0 | W1] is the same 64 numbers; formed per tile, not kept across the loop: registers)
                double z1d[8];
                float z0c[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = (j < 4 ? 4 * cfirst : 4 * csecond) + (j & 3);
                    float q0 = 0.f, q1 = 0.f;
                    for (int kk = 0; kk < F.cin; ++kk) {
                        q0 = fmaf(F.x_const, F.W0[kk * 64 + c], q0);
                        q1 = fmaf(F.x_const, F.W0[kk * 64 + kBH + c], q1);
                    }
                    z0c[j] = q0;
                    z1d[j] = (double)q1;
                }
                const int crow = has ? (int)cnt[perm[slot]] : 0;
                double accd[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) accd[j] = 0.0;
#define B2_TRIP_FRONT(R, TT)                                                                                             \
                {                                                                                                      \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
                        const double ad = (double)__int_as_float(e == 0 ? B2_QB(R.x, 0) : e == 1 ? B2_QB(R.x, 1) : e == 2 ? B2_QB(R.x, 2) : B2_QB(R.x, 3)); \
                        if (4 * (TT) + e < crow) {                                                                     \
                            _Pragma("unroll") for (int j = 0; j < 8; ++j) accd[j] = fma(ad, z1d[j], accd[j]);          \
                        }                                                                                              \
                    }                                                                                                  \
                }
                B2_WALK(B2_TRIP_FRONT)
#undef B2_TRIP_FRONT
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    double d = (double)z0c[j] + accd[j];
                    if (F.bias0) d += (double)F.bias0[(j < 4 ? 4 * cfirst : 4 * csecond) + (j & 3)];
                    o[j] = big_act((float)d, F.act0);
                }
                const float4 oA = make_float4(o[0], o[1], o[2], o[3]), oB = make_float4(o[4], o[5], o[6], o[7]);
                B2_STAGE_TO_OPERAND(oA, oB)
            }
        }
    } else {
        // explicit input features: the caller ran layer index 0 layer by layer; its output H [num_nodes][32] in operand layout
#pragma unroll 1
        for (int k = 0; k < TILES; ++k) {
            const int mslot = (wave + kB2Waves * k) * 16 + mr;
            float hh[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) hh[s] = 0.f;
            if (mslot < ng) {
                const float* hr = a.Zin + (size_t)(n0 + (int)perm[mslot]) * kBH;
#pragma unroll
                for (int s = 0; s < 8; ++s) hh[s] = hr[4 * s + mq];
            }
            B2_SET_H(k, hh)
        }
    }
    // (the last layer's - width 1 - z0 / z1 of this lane's row end up in alo[k].x / .y: the lo sums are dead by then)
    B2_STAMP(2);  // layer 0 (or H of layer 0 read from global memory)
    for (int i = 0; i < a.num_hidden; ++i) {
        const BigLayer& L = a.layers[i];
        const bool last = i == a.num_hidden - 1;
        const float* Wz = i == 0 ? a.first.W1 : a.layers[i - 1].Wnext;  // the transform in front of this aggregation
        const bool f64 = i == 0;  // layer index 1: every chain in double, rounded once (include/dgcn.h)
        float bcol[8];
        // FUSE (twelve tiles per wave): Z0's lo half and the lo half of H' are formed in S2 behind the tile's walk, beside other
        // waves' walks, where the matrix pipes idle; what is left of the MFMA-only phase S3 is Z1 hi and Z0's hi half.  MC1500:
        // S3 248 -> 148 us, S2 206 -> 263 us per launch, 1 076 -> 1 036 us.  (Eight tiles per wave - two record groups in flight -
        // lose by it: ER(1 000, 0.01) 750 -> 778 us.  The next layer's Z1 lo formed in S4's epilogue into the lo sums' registers,
        // so that S1 only stores: those registers then live through S4's walk, H spills at twelve tiles - 104 registers -, and
        // at eight it costs another 15 us: dropped.)
        constexpr bool FUSE = TILES == 12;
        // -------- S1: Z1 lo = H.W1[:, 0:16] -> LDS (every gather of the previous aggregation is behind a barrier)
        {
            b2_load_bcol(Wz, 2, f64, bcol);
#pragma unroll 1
            for (int k = 0; k < TILES; k += 2) {  // two tiles at a time: two independent MFMA chains
                const int t = wave + kB2Waves * k;
                if (t < tiles) {
                    B2_LAUNDER
                    const int k1 = min(k + 1, TILES - 1);  // (the second tile may not exist: its product is not stored then)
                    float h0[8], h1[8];
                    B2_GET_H(k, h0)
                    B2_GET_H(k1, h1)
                    float4 o0, o1;
                    b2_mma2(bcol, h0, h1, f64, o0, o1);
                    const int mslot = t * 16 + mr, mslot1 = mslot + kB2Waves * 16;
                    if (mslot < nr) *reinterpret_cast<float4*>(bufH + (int)perm[mslot] * 16 + 4 * mq) = o0;
                    if (k + 1 < TILES && mslot1 < nr) *reinterpret_cast<float4*>(bufH + (int)perm[mslot1] * 16 + 4 * mq) = o1;
                }
            }
        }
        B2_STAMP(3);  // S1 (sum over the layers)
        __syncthreads();
        B2_STAMP(4);
        // -------- S2: the lo halves of the row sums (their registers hold nothing between S3 and here: written whole, so that
        // the compiler sees them dead across S4 and S1, where the walk's and the staging's registers are needed)
        if constexpr (!STASH) {
#pragma unroll
            for (int j = 0; j < 4; ++j) alo[j] = (vtile)(0.f);
        }
        float bz0[8];
        if (FUSE) b2_load_bcol(Wz, 0, f64, bz0);
        B2_FIRST_GROUP
#pragma unroll 1
        for (int k = 0; k < TILES; ++k) {
            if (wave + kB2Waves * k < tiles) {
                B2_TILE_HEAD
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                B2_WALK(B2_TRIP)
                if (FUSE) {
                    // Z0's lo half and the lo half of H' = act(Z0 + L.Z1 + b) right here: the tile's lo sums are complete
                    float hh[8];
                    B2_GET_H(k, hh)
                    const float4 z0a = b2_mma(bz0, hh, f64);
                    *reinterpret_cast<float4*>(stg + mr * kBH + ((mq ^ (mr & 7)) << 2)) = z0a;  // MFMA layout (row mr, chunk mq) ->
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_wave_barrier();
                    const float4 yA = *reinterpret_cast<const float4*>(stg + s16 * kBH + ((cfirst ^ (s16 & 7)) << 2));  // aggregation layout
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    __builtin_amdgcn_wave_barrier();
                    acc = make_float4(yA.x + acc.x, yA.y + acc.y, yA.z + acc.z, yA.w + acc.w);
                    if (L.bias) {
                        const float4 biasA = *reinterpret_cast<const float4*>(L.bias + 4 * cfirst);
                        acc.x += biasA.x; acc.y += biasA.y; acc.z += biasA.z; acc.w += biasA.w;
                    }
                    acc.x = big_act(acc.x, L.act); acc.y = big_act(acc.y, L.act); acc.z = big_act(acc.z, L.act); acc.w = big_act(acc.w, L.act);
                }
                B2_LO_PUT(k, acc)
            }
        }
        b2_load_bcol(Wz, 3, f64, bcol);  // (requested here: they land while this wave waits at the barrier)
        float bz1[8];
        b2_load_bcol(Wz, 1, f64, bz1);
        if (!FUSE) b2_load_bcol(Wz, 0, f64, bz0);
        B2_STAMP(5);  // S2
        __syncthreads();  // every lo gather has read Z1h
        B2_STAMP(6);
        // -------- S3: Z1 hi = H.W1[:, 16:32] -> LDS; Z0's hi half (FUSE: its lo half and the lo half of H' were formed in S2).
        // H is dead afterwards: its registers carry Z0's hi half [0..3] and the lo half of H' [4..7] (aggregation layout) to S4.
#pragma unroll 1
        for (int k = 0; k < TILES; ++k) {
            const int t = wave + kB2Waves * k;
            if (t < tiles) {
                B2_LAUNDER
                float hh[8];
                B2_GET_H(k, hh)
                float4 o, z0a = make_float4(0.f, 0.f, 0.f, 0.f), z0b;
                if (FUSE) b2_mma2b(bcol, bz1, hh, f64, o, z0b);
                else b2_mma3(bcol, bz0, bz1, hh, f64, o, z0a, z0b);
                const int mslot = t * 16 + mr;
                if (mslot < nr) *reinterpret_cast<float4*>(bufH + (int)perm[mslot] * 16 + 4 * mq) = o;
                // MFMA output layout (row mr: chunk mq of z0a, chunk 4 + mq of z0b) -> aggregation layout
                if (!FUSE) *reinterpret_cast<float4*>(stg + mr * kBH + ((mq ^ (mr & 7)) << 2)) = z0a;
                *reinterpret_cast<float4*>(stg + mr * kBH + (((4 + mq) ^ (mr & 7)) << 2)) = z0b;
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
                float4 yA = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!FUSE) yA = *reinterpret_cast<const float4*>(stg + s16 * kBH + ((cfirst ^ (s16 & 7)) << 2));
                const float4 yB = *reinterpret_cast<const float4*>(stg + s16 * kBH + ((csecond ^ (s16 & 7)) << 2));
                __builtin_amdgcn_s_waitcnt(0xC07F);
                __builtin_amdgcn_wave_barrier();
                float4 lo = make_float4(0.f, 0.f, 0.f, 0.f);
                B2_LO_GET(k, lo)
                float4 oA = lo;  // FUSE: the lo half of H' already
                if (!FUSE) {
                    oA = make_float4(yA.x + lo.x, yA.y + lo.y, yA.z + lo.z, yA.w + lo.w);
                    if (L.bias) {
                        const float4 biasA = *reinterpret_cast<const float4*>(L.bias + 4 * cfirst);
                        oA.x += biasA.x; oA.y += biasA.y; oA.z += biasA.z; oA.w += biasA.w;
                    }
                    oA.x = big_act(oA.x, L.act); oA.y = big_act(oA.y, L.act); oA.z = big_act(oA.z, L.act); oA.w = big_act(oA.w, L.act);
                }
                hh[0] = yB.x; hh[1] = yB.y; hh[2] = yB.z; hh[3] = yB.w;
                hh[4] = oA.x; hh[5] = oA.y; hh[6] = oA.z; hh[7] = oA.w;
                B2_SET_H(k, hh)
            }
        }
        B2_STAMP(7);  // S3
        __syncthreads();
        B2_STAMP(8);
        // -------- S4: the hi halves of the row sums, the hi half of H', H' into the operand layout
        B2_FIRST_GROUP
#pragma unroll 1
        for (int k = 0; k < TILES; ++k) {
            if (wave + kB2Waves * k < tiles) {
                B2_TILE_HEAD
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                B2_WALK(B2_TRIP)
                float hh[8];
                B2_GET_H(k, hh)
                const float4 oA = make_float4(hh[4], hh[5], hh[6], hh[7]);
                float4 oB = make_float4(hh[0] + acc.x, hh[1] + acc.y, hh[2] + acc.z, hh[3] + acc.w);
                if (L.bias) {
                    const float4 biasB = *reinterpret_cast<const float4*>(L.bias + 4 * csecond);
                    oB.x += biasB.x; oB.y += biasB.y; oB.z += biasB.z; oB.w += biasB.w;
                }
                oB.x = big_act(oB.x, L.act); oB.y = big_act(oB.y, L.act); oB.z = big_act(oB.z, L.act); oB.w = big_act(oB.w, L.act);
                if (last) {
                    // the last layer (32 -> 1): z = H'.[w0 | w1], two fmaf chains over k = 0..31 round the row's four lanes
                    // (chunk c of the row sits in lane c & 3, as its lo or its hi float4)
                    const float* Wl = a.Wlast;
                    float q0 = 0.f, q1 = 0.f;
#define B2_LAST_STEP(C)                                                                                                  \
                    {                                                                                                  \
                        const float4 o = ((C) < 4) ? oA : oB;                                                          \
                        float t0 = q0, t1 = q1;                                                                        \
                        t0 = fmaf(o.x, Wl[(4 * (C) + 0) * 2], t0); t1 = fmaf(o.x, Wl[(4 * (C) + 0) * 2 + 1], t1);      \
                        t0 = fmaf(o.y, Wl[(4 * (C) + 1) * 2], t0); t1 = fmaf(o.y, Wl[(4 * (C) + 1) * 2 + 1], t1);      \
                        t0 = fmaf(o.z, Wl[(4 * (C) + 2) * 2], t0); t1 = fmaf(o.z, Wl[(4 * (C) + 2) * 2 + 1], t1);      \
                        t0 = fmaf(o.w, Wl[(4 * (C) + 3) * 2], t0); t1 = fmaf(o.w, Wl[(4 * (C) + 3) * 2 + 1], t1);      \
                        q0 = B2_QBF(t0, (C) & 3);                                                                      \
                        q1 = B2_QBF(t1, (C) & 3);                                                                      \
                    }
                    B2_LAST_STEP(0) B2_LAST_STEP(1) B2_LAST_STEP(2) B2_LAST_STEP(3)
                    B2_LAST_STEP(4) B2_LAST_STEP(5) B2_LAST_STEP(6) B2_LAST_STEP(7)
#undef B2_LAST_STEP
                    const float4 zq01 = make_float4(q0, q1, 0.f, 0.f);
                    B2_LO_PUT(k, zq01)
                } else {
                    B2_STAGE_TO_OPERAND(oA, oB)
                }
            }
        }
        B2_STAMP(9);  // S4
        if (!last) __syncthreads();  // every hi gather has read Z1h: the next layer's S1 writes it
        B2_STAMP(10);
    }
    // -------- the last layer's aggregation at width 1: z1 of the whole graph as a float array over the Z1h space
    __syncthreads();
    float* zl = bufH;
#pragma unroll
    for (int k = 0; k < TILES; ++k) {
        const int slot = (wave + kB2Waves * k) * 16 + s16;
        float4 zq01 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wave + kB2Waves * k < tiles) B2_LO_GET(k, zq01)
        if (slot < nr && kq4 == 0) zl[perm[slot]] = zq01.y;
    }
    if (threadIdx.x == 0) zl[a.max_nodes] = 0.f;  // the neutral record's neighbour
    __syncthreads();
    B2_FIRST_GROUP
#pragma unroll 1
    for (int k = 0; k < TILES; ++k) {
        if (wave + kB2Waves * k < tiles) {
            B2_TILE_HEAD
            float accs = 0.f;
            float4 zq01 = make_float4(0.f, 0.f, 0.f, 0.f);
            B2_LO_GET(k, zq01)
            const float z0k = zq01.x;
#define B2_TRIP_TAIL(R, TT)                                                                                              \
            {                                                                                                          \
                float zs[4], av[4];                                                                                    \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                        \
                    const unsigned w = (unsigned)(e == 0 ? B2_QB(R.y, 0) : e == 1 ? B2_QB(R.y, 1) : e == 2 ? B2_QB(R.y, 2) : B2_QB(R.y, 3)); \
                    av[e] = __int_as_float(e == 0 ? B2_QB(R.x, 0) : e == 1 ? B2_QB(R.x, 1) : e == 2 ? B2_QB(R.x, 2) : B2_QB(R.x, 3)); \
                    zs[e] = zl[w >> 6];                                                                                \
                }                                                                                                      \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) accs = fmaf(av[e], zs[e], accs);                         \
            }
            B2_WALK(B2_TRIP_TAIL)
#undef B2_TRIP_TAIL
            float o = z0k + accs;
            if (a.bias_last) o += a.bias_last[0];
            o = big_act(o, a.act_last);
            if (has && kq4 == 0 && trips > 0) {
                const int v = (int)perm[slot];
                a.scores[n0 + v] = o;
                if (a.do_lgs) {
                    double p = (double)o;
                    if (a.predict_mwis && a.weights) p *= a.weights[n0 + v];
                    reinterpret_cast<double*>(b2_lds + big_lgs_base(a.max_nodes))[v] = p;
                }
            }
        }
    }
    if (a.do_lgs) {
        // -------- the local greedy search (heuristics.py:77-116): k_lgs's rounds on priorities, state bytes and row offsets in LDS
        // (the votes through acc64[3]: ockl's workgroup reductions would bring static LDS, and Z1h sits at LDS offset 0)
        double* pr = reinterpret_cast<double*>(b2_lds + big_lgs_base(a.max_nodes));
        double* red = pr + a.max_nodes;                                                // [kB2Block]
        unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(red + kB2Block);  // [4]
        int* rol = reinterpret_cast<int*>(acc64 + 4);                                   // [max_nodes + 1]
        uint8_t* st = reinterpret_cast<uint8_t*>(rol + ((a.max_nodes + 1 + 3) & ~3));
        // (st and nw have a slot [ng] - max_nodes + 1 bytes rounded up, as b2_lgs_lds() sizes them: where a column outside the graph points, state 3)
        uint8_t* nw = st + ((a.max_nodes + 16) & ~15);
        // the graph's columns as 16-bit local ids in what is left of the Z1h space, when they fit: the rounds then read LDS only
        uint16_t* cl = reinterpret_cast<uint16_t*>(nw + ((a.max_nodes + 16) & ~15));
        const int cl_cap = (int)(((size_t)a.max_nodes * 64 - (size_t)(reinterpret_cast<unsigned char*>(cl) - b2_lds)) / 2);
        const int e0 = a.arow[n0], e1 = a.arow[n0 + ng];
        const bool cols_lds = (e1 - e0) <= cl_cap;
        __syncthreads();  // every priority is written, every walk over z1 done
        int bad = 0;
        for (int v = threadIdx.x; v < ng; v += kB2Block) {
            // (residual step) who takes part: the vertices that were undecided when the launch began (read again: the Z1h space held
            // the alive bytes only until the first layer); a decided vertex reports score 0 and keeps state 3 = "not part of this"
            const bool part = !resid || a.state[n0 + v] == 0;
            if (part) {
                const double p = pr[v];
                bad |= p != p;
            } else {
                a.scores[n0 + v] = 0.f;
                pr[v] = 0.0;
            }
            st[v] = part ? 0 : 3;
            nw[v] = 0;
        }
        for (int v = threadIdx.x; v <= ng; v += kB2Block) rol[v] = a.arow[n0 + v];
        if (cols_lds) {
            for (int base = e0 + threadIdx.x; base < e1; base += kB2Block * 4) {  // four loads in flight per thread
                int c[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = (base + i * kB2Block < e1) ? a.acol[base + i * kB2Block] : n0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (base + i * kB2Block < e1) {  // (a column outside the graph - BAD_COLUMN is reported with the batch - reads the sentinel)
                        const int u = c[i] - n0;
                        cl[base + i * kB2Block - e0] = (uint16_t)((unsigned)u < (unsigned)ng ? u : ng);
                    }
            }
        }
        if (threadIdx.x == 0) { acc64[3] = 0; st[ng] = 3; nw[ng] = 0; }
        __syncthreads();
        if (bad) acc64[3] = 1;
        __syncthreads();
        if (acc64[3] != 0) {  // the reference would spin forever on a NaN priority: report instead
            if (threadIdx.x == 0) {
                atomicOr(a.status, fault | DGCN_FAULT_NAN_PRIORITY);
                if (a.rounds) a.rounds[g] = -1;
                if (a.totals) a.totals[g] = 0.0;
                if (a.active) a.active[g] = 0;
            }
            if (!resid) for (int v = threadIdx.x; v < ng; v += kB2Block) a.state[n0 + v] = 0;
            return;
        }
        __syncthreads();  // (acc64[3] is the rounds' vote word from here on)
        if (resid && threadIdx.x == 0) {
            if (a.progress) atomicAdd(a.progress, 1);
            if (a.tail_word) atomicMax(a.tail_word, a.tail_tag | (unsigned long long)(unsigned)nr);
        }
        if (resid && a.greedy_mode == 2) {  // the rollout: priorities out; candidates, completions and the pick below (or, beyond sixteen candidates, general.hip's launches)
            for (int v = threadIdx.x; v < ng; v += kB2Block) a.prio_out[n0 + v] = st[v] == 0 ? pr[v] : 0.0;
            if (threadIdx.x == 0 && a.active) a.active[g] = 1;
            if (fault) atomicOr(a.status, fault);
            if (a.cid) {  // the candidates right here: no k_res_cand launch (cand_select.h; the staging tiles' space is free)
                static_assert(kCandPer * kB2Block >= kB2MaxNodes, "a thread holds its vertices' priorities in registers");
                int32_t* cid = a.cid + (size_t)g * kCandMaxBeam;
                double pv[kCandPer];
                unsigned have = 0u;
                const int per = (ng + kB2Block - 1) / kB2Block;
#pragma unroll
                for (int i = 0; i < kCandPer; ++i) {
                    const int v = (int)threadIdx.x + i * kB2Block;
                    pv[i] = 0.0;
                    if (i < per && v < ng && st[v] == 0) { pv[i] = pr[v]; have |= 1u << i; }
                }
                if (a.roll_off) {
                    // the whole step here: candidates (list mirrored in LDS), the completions of all of them at once, the pick
                    // (the host has made sure every graph's columns are in LDS: cols_lds)
                    int32_t* cidl = reinterpret_cast<int32_t*>(b2_lds + a.roll_off);
                    if (threadIdx.x < kCandMaxBeam) cidl[threadIdx.x] = -1;
                    __syncthreads();
                    cand_select<kB2Block>(pv, have, per, min(a.beam, kCandMaxBeam), cid, b2_lds + a.lds_stage_off, cidl);
                    if (!a.by_priority)  // (the completions go by weight: mwis_gdpg_call.py:640)
                        for (int v = threadIdx.x; v < ng; v += kB2Block) pr[v] = a.weights[n0 + v];
                    __syncthreads();
                    RollArgs r;
                    r.ng = ng; r.n0 = n0; r.e0 = e0;
                    r.key = pr; r.st = st; r.rol = rol; r.cl = cl; r.cidl = cidl; r.beam = a.beam;
                    r.extra = reinterpret_cast<unsigned char*>(cidl + kCandMaxBeam);
                    r.max_nodes = a.max_nodes;
                    r.wl = a.by_priority ? nullptr : pr; r.weights = a.weights; r.state = a.state; r.rounds = a.rounds; r.totals = a.totals;
                    rollout_bits<kB2Block>(r, g);
                    return;
                }
                __syncthreads();
                cand_select<kB2Block>(pv, have, per, min(a.beam, kCandMaxBeam), cid, b2_lds + a.lds_stage_off);
            }
            return;
        }
        if (resid && a.greedy_mode == 1) {
            // solve_mwis_cit: the best-priority undecided vertex joins (np.argmax: lowest index among equals), its neighbours leave
            double bp = 0.0;
            int bv = -1;
            for (int v = threadIdx.x; v < ng; v += kB2Block) {
                if (st[v] != 0) continue;
                const double p = pr[v];
                if (bv < 0 || p > bp) { bp = p; bv = v; }  // ascending v per thread: the first maximum stays
            }
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double op = __shfl_xor(bp, off);
                const int ov = __shfl_xor(bv, off);
                if (ov >= 0 && (bv < 0 || op > bp || (op == bp && ov < bv))) { bp = op; bv = ov; }
            }
            int* ri = reinterpret_cast<int*>(red + 16);
            if (lane == 0) { red[wave] = bp; ri[wave] = bv; }
            __syncthreads();
            bp = red[0]; bv = ri[0];
#pragma unroll
            for (int w = 1; w < kB2Waves; ++w) {
                const double op = red[w];
                const int ov = ri[w];
                if (ov >= 0 && (bv < 0 || op > bp || (op == bp && ov < bv))) { bp = op; bv = ov; }
            }
            if (bv >= 0) {
                for (int j = rol[bv] + (int)threadIdx.x; j < rol[bv + 1]; j += kB2Block) {
                    const int u = a.acol[j] - n0;
                    if ((unsigned)u < (unsigned)ng && u != bv && st[u] == 0) a.state[n0 + u] = 2;
                }
                if (threadIdx.x == 0) {
                    a.state[n0 + bv] = 1;
                    if (a.rounds) a.rounds[g] = 1;
                    if (a.totals) a.totals[g] = a.weights ? a.weights[n0 + bv] : bp;
                }
            }
            if (fault) atomicOr(a.status, fault);
            return;
        }
        LgsArgs la = {};
        la.col_idx = a.acol;
        la.rounds = a.rounds;
        la.max_rounds = resid ? a.max_rounds : 0;
        la.init_state = resid ? a.state : nullptr;  // (only its being there matters: the rounds count who takes part from `st`)
        if (cols_lds && la.max_rounds <= 0 && a.ahead_rounds) {
            // a whole search: the rounds on ahead lists (lgs_rounds.h; the counts in the reduction array's space, free until the totals)
            static_assert(kB2Block * 4 >= kB2MaxNodes, "a 16-bit count per vertex in the reduction array");
            const int rounds = lgs_rounds_ahead<kB2Block, true>(ng, e0, pr, st, nw, cl, rol, reinterpret_cast<uint16_t*>(red), acc64);
            if (threadIdx.x == 0 && a.rounds) a.rounds[g] = rounds;
            __syncthreads();
        } else
        if (cols_lds) lgs_rounds<1, false, true, kB2Block, true>(la, g, n0, ng, e0, pr, st, nw, cl, acc64, rol);
        else lgs_rounds<1, false, false, kB2Block, true>(la, g, n0, ng, e0, pr, st, nw, nullptr, acc64, rol);
        {
            double part = 0.0;
            for (int v = threadIdx.x; v < ng; v += kB2Block) {
                const uint8_t s1 = st[v];
                if (resid) {
                    if (s1 == 1 || s1 == 2) {  // (3: decided before this step, 0: still undecided after max_rounds rounds)
                        if (a.totals && s1 == 1) part += a.weights ? a.weights[n0 + v] : pr[v];
                        a.state[n0 + v] = s1;
                    }
                } else {
                    if (a.totals && s1 == 1) part += a.weights ? a.weights[n0 + v] : pr[v];
                    a.state[n0 + v] = s1;
                }
            }
            red[threadIdx.x] = part;
        }
        if (a.totals) {
            __syncthreads();
            for (int off = kB2Block / 2; off > 0; off >>= 1) {
                if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
                __syncthreads();
            }
            if (threadIdx.x == 0) a.totals[g] = red[0];
        }
    }
    B2_STAMP(11);  // last layer's width-1 walk, search, totals
#ifdef DGCN_DIAG
    if (a.stamps && threadIdx.x == 0) {
#pragma unroll
        for (int _i = 0; _i < 12; ++_i) a.stamps[(size_t)g * 16 + _i] = b2_acc[_i];
        a.stamps[(size_t)g * 16 + 13] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#undef B2_TRIP
#undef B2_LO_GET
#undef B2_LO_PUT
#undef B2_SET4
#undef B2_GET4
#undef B2_SET_H
#undef B2_GET_H
#undef B2_STAGE_TO_OPERAND
#undef B2_FIRST_GROUP
#undef B2_WALK
#undef B2_PREFETCH
#undef B2_TILE_HEAD
#undef B2_LAUNDER
#undef B2_QBF
#undef B2_QB
    if (fault) atomicOr(a.status, fault);
}

// ---------------------------------------------------------------------------------------------------------------------
int spmm_f64acc_dispatch(const DgcnCsr* S, const int32_t* graph_ptr, int B, int max_nodes, const float* Z, int ldz, int C,
                         const float* Y0, int ldy0, const float* bias, int act, float* Y, int ldy, hipStream_t s);
int transform_dispatch(const float* H, int ldh, float h_const, int rows, int cin, const float* W, int ctot, float* Z,
                       int ldz, hipStream_t s);

static size_t b2_256(size_t x) { return (x + 255) & ~(size_t)255; }

// option "big2": 0 = no k_big2 (layer by layer), 1 = also for graphs k_big / k_fused would take (tests)
static int b2_env() { return opt(OPT_BIG2); }

static int b2_rec_cap(const DgcnBatch* b) {
    return ((b->max_graph_edges + b->max_nodes + 2 + 16 + 15) & ~15) + ((20 * b->max_nodes + 448 + 15) & ~15);
}

static size_t b2_lds_bytes(int max_nodes, int* cnt_off, int* perm_off, int* stage_off, int* tab_off) {
    size_t off = (size_t)max_nodes * 64 + 64;  // Z1h + the zero row
    *stage_off = (int)off;
    off += (size_t)kB2Waves * 2048;            // a 16 x 32 float tile per wave (P0: the count histogram, the d^-1/2 array)
    *cnt_off = (int)off;
    off += ((size_t)max_nodes * 2 + 15) & ~(size_t)15;
    *perm_off = (int)off;
    off += ((size_t)max_nodes * 2 + 15) & ~(size_t)15;
    *tab_off = (int)off;
    off += kB2MaxTiles * 8;
    return off;
}

static int b2_ahead_rounds() {  // option "wide_ahead" = 0: lgs_rounds.h's three-phase rounds (the tests' witness)
    const bool off = opt(OPT_WIDE_AHEAD) == 0;
    return off ? 0 : 1;
}

static size_t b2_lgs_lds(int max_nodes) {
    const size_t pad = (size_t)((max_nodes + 16) & ~15);  // (state / winner bytes: max_nodes + 1 - the sentinel slot [ng] -, rounded up; the kernel carves the same)
    return big_lgs_base(max_nodes) + (size_t)max_nodes * 8 + kB2Block * 8 + 4 * 8 + (size_t)((max_nodes + 1 + 3) & ~3) * 4 + 2 * pad + 16;
}

// 1 = a deep [I, L] stack F -> 32 -> .. -> 32 -> 1 on graphs of at most 1 920 vertices that k_big does not take
int big_takes(const DgcnBatch* b, const DgcnModel* m);
int big2_takes(const DgcnBatch* b, const DgcnModel* m) {
    if (b2_env() == 0) return 0;
    if (!b || !m || !m->layers_host || m->num_supports != 2 || m->num_layers < 3 || m->num_layers - 2 > kBigMaxLayers) return 0;
    if (b->max_nodes <= 0 || b->max_nodes > kB2MaxNodes) return 0;
    if (b2_env() != 1 && big_takes(b, m)) return 0;
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

static size_t b2_stash_bytes(const DgcnBatch* b) {  // the lo sums' slice per graph (more than twelve tiles per wave only)
    const size_t mn = (size_t)((std::max(b->max_nodes, 16) + 15) & ~15);
    return mn / 16 > 12 * kB2Waves ? mn * 64 : 0;
}

size_t big2_workspace(const DgcnBatch* b, const DgcnModel* m) {
    if (!big2_takes(b, m)) return 0;
    const size_t B = (size_t)std::max(b->num_graphs, 1);
    return 256 + b2_256(B * (size_t)b2_rec_cap(b) * 8) + b2_256(B * b2_stash_bytes(b));
}

static void b2_carve(BigArgs& a, const DgcnBatch* b, void* bws) {
    const size_t B = (size_t)std::max(b->num_graphs, 1);
    char* w = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(bws) + 255) & ~(uintptr_t)255);
    a.rec = reinterpret_cast<uint2*>(w);
    a.stash = reinterpret_cast<float4*>(w + b2_256(B * (size_t)b2_rec_cap(b) * 8));
}

template <int TILES, bool RESID = false>
static int big2_launch_t(BigArgs& a, int B, size_t lds, const char* family, hipStream_t s) {
    if (lds > 64 * 1024) {
        static std::atomic<int> reserved[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!reserved[dev & 63].load(std::memory_order_relaxed)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_big2<TILES, RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return fail(DGCN_ERR_LAUNCH, "k_big2: cannot reserve %zu bytes of LDS", lds);
            reserved[dev & 63].store(1, std::memory_order_relaxed);
        }
    }
    TimedLaunch t(family, s);
    DGCN_LAUNCH(t, (k_big2<TILES, RESID>), dim3((unsigned)B), dim3(kB2Block), lds, s, a);
    return check_launch("k_big2");
}

static int big2_launch(BigArgs& a, int B, size_t lds, const char* family, hipStream_t s) {
    if (lds > 160 * 1024) return fail(DGCN_ERR_UNSUPPORTED, "k_big2: %zu bytes of LDS", lds);
#ifdef DGCN_DIAG
    a.stamps = reinterpret_cast<unsigned long long*>(static_cast<uintptr_t>(opt64(OPT_DIAG_STAMPS)));
#endif
    const int per_wave = (a.max_nodes / 16 + kB2Waves - 1) / kB2Waves;  // tiles a wave owns at most
    if (a.residual) {
        if (per_wave <= 8) return big2_launch_t<8, true>(a, B, lds, family, s);
        if (per_wave <= 12) return big2_launch_t<12, true>(a, B, lds, family, s);
        return big2_launch_t<15, true>(a, B, lds, family, s);
    }
    if (per_wave <= 8) return big2_launch_t<8>(a, B, lds, family, s);
    if (per_wave <= 12) return big2_launch_t<12>(a, B, lds, family, s);
    return big2_launch_t<15>(a, B, lds, family, s);
}

static void big2_fill_model(BigArgs& a, const DgcnModel* m, float x_const) {
    const int Lc = m->num_layers;
    const DgcnLayer& L0 = m->layers_host[0];
    const DgcnLayer& L1 = m->layers_host[1];
    const DgcnLayer& LL = m->layers_host[Lc - 1];
    a.first.W0 = L0.weights; a.first.bias0 = L0.bias; a.first.W1 = L1.weights; a.first.x_const = x_const;
    a.first.cin = L0.in_dim; a.first.act0 = L0.act;
    a.Wlast = LL.weights; a.bias_last = LL.bias; a.act_last = LL.act;
    a.num_hidden = Lc - 2;
    for (int l = 1; l <= Lc - 2; ++l) {
        const DgcnLayer& L = m->layers_host[l];
        a.layers[l - 1].bias = L.bias;
        a.layers[l - 1].act = L.act;
        a.layers[l - 1].Wnext = l < Lc - 2 ? m->layers_host[l + 1].weights : nullptr;
    }
}

// The forward pass in one launch (constant input features), or - explicit features - layer index 0 by the layer-by-layer
// kernels exactly as layered_forward runs it (transform, aggregation with the row chains in double), then everything else.
// `lws`: dgcn_gcn_forward_workspace(b, m, 0) bytes (Z twice, H), `bws`: big2_workspace(b, m) bytes.
int big2_forward(const DgcnBatch* b, const DgcnCsr* lap, const DgcnModel* m, const float* X, float x_const, float* scores,
                 void* lws, void* bws, int32_t* status, hipStream_t s) {
    const size_t n = (size_t)b->num_nodes;
    char* w0 = reinterpret_cast<char*>(lws);
    const size_t zsz = b2_256(n * 2 * kBH * sizeof(float));
    float* Zbuf = reinterpret_cast<float*>(w0);
    float* Hbuf = reinterpret_cast<float*>(w0 + 2 * zsz);
    const DgcnLayer& L0 = m->layers_host[0];
    const bool front = X == nullptr && L0.in_dim <= 64;
    if (!front) {
        int rc = transform_dispatch(X, L0.in_dim, x_const, b->num_nodes, L0.in_dim, L0.weights, 2 * kBH, Zbuf, 2 * kBH, s);
        if (rc) return rc;
        rc = spmm_f64acc_dispatch(lap, b->graph_ptr, b->num_graphs, b->max_nodes, Zbuf + kBH, 2 * kBH, kBH, Zbuf, 2 * kBH, L0.bias, L0.act,
                                  Hbuf, kBH, s);
        if (rc) return rc;
    }
    BigArgs a = {};
    a.graph_ptr = b->graph_ptr;
    a.lrow = lap->row_ptr; a.lcol = lap->col_idx; a.lval = lap->values;
    a.Zin = Hbuf;
    b2_carve(a, b, bws);
    a.status = status;
    a.front = front ? 1 : 0;
    a.scores = scores;
    a.rec_cap = b2_rec_cap(b);
    a.max_nodes = (std::max(b->max_nodes, 16) + 15) & ~15;
    big2_fill_model(a, m, x_const);
    const size_t lds = b2_lds_bytes(a.max_nodes, &a.lds_cnt_off, &a.lds_perm_off, &a.lds_stage_off, &a.lds_tab_off);
    return big2_launch(a, b->num_graphs, lds, "big_forward", s);
}

// 1 = dgcn_solve_batch's whole path in ONE launch: adjacency in, set out (constant input features, k_big2's shapes)
int big2_solve_takes(const DgcnBatch* b, const DgcnModel* m, const float* X) {
    const int solve = opt(OPT_BIG_SOLVE);
    if (solve == 0) return 0;
    return !X && big2_takes(b, m) && m->layers_host[0].in_dim <= 64;
}

int big2_solve(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, float x_const, const double* weights,
               int32_t predict_mwis, float* scores, uint8_t* state, int32_t* rounds, double* totals, int32_t* status, void* bws,
               hipStream_t s) {
    BigArgs a = {};
    a.graph_ptr = b->graph_ptr;
    a.arow = b->row_ptr; a.acol = b->col_idx; a.dinv = dinv_table; a.table_len = table_len;
    b2_carve(a, b, bws);
    a.status = status;
    a.rec_cap = b2_rec_cap(b);
    a.max_nodes = (std::max(b->max_nodes, 16) + 15) & ~15;
    a.front = 1;
    a.scores = scores;
    a.do_lgs = 1; a.predict_mwis = predict_mwis; a.ahead_rounds = b2_ahead_rounds();
    a.weights = weights; a.state = state; a.rounds = rounds; a.totals = totals;
    big2_fill_model(a, m, x_const);
    const size_t lds = std::max(b2_lds_bytes(a.max_nodes, &a.lds_cnt_off, &a.lds_perm_off, &a.lds_stage_off, &a.lds_tab_off),
                                b2_lgs_lds(a.max_nodes));
    return big2_launch(a, b->num_graphs, lds, "big_solve", s);
}

// One step of dgcn_solve_residual_batch in ONE launch on k_big2's shapes (constant input features): big.hip's big_residual for
// graphs of 977 .. 1 920 vertices.  option "big_residual" = 0: the compaction launches + k_big2 + k_lgs instead.
int big2_residual_takes(const DgcnBatch* b, const DgcnModel* m, const float* X, int32_t feature_mode, int32_t options) {
    const int on = opt(OPT_BIG_RESIDUAL);
    if (on == 0 || feature_mode != 0 || (options & DGCN_RESIDUAL_SCORES_GIVEN)) return 0;
    return big2_solve_takes(b, m, X);
}

int big2_residual(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, float x_const, const double* weights,
                  int32_t predict_mwis, int32_t greedy_mode, int32_t max_rounds, float* scores, uint8_t* state, int32_t* rounds,
                  double* totals, int32_t* progress, int32_t* status, double* prio, int32_t* active, int32_t* cid, int32_t beam,
                  int32_t by_priority, int32_t* whole_step, unsigned long long* tail_word, unsigned long long tail_tag, void* bws,
                  hipStream_t s) {
    if (whole_step) *whole_step = 0;
    BigArgs a = {};
    a.graph_ptr = b->graph_ptr;
    a.arow = b->row_ptr; a.acol = b->col_idx; a.dinv = dinv_table; a.table_len = table_len;
    b2_carve(a, b, bws);
    a.status = status;
    a.rec_cap = b2_rec_cap(b);
    a.max_nodes = (std::max(b->max_nodes, 16) + 15) & ~15;
    a.front = 1;
    a.scores = scores;
    a.do_lgs = 1; a.predict_mwis = predict_mwis; a.ahead_rounds = b2_ahead_rounds();
    a.weights = weights; a.state = state; a.rounds = rounds; a.totals = totals;
    a.residual = 1; a.greedy_mode = greedy_mode; a.max_rounds = max_rounds;
    a.progress = progress; a.tail_word = tail_word; a.tail_tag = tail_tag;
    a.prio_out = greedy_mode == 2 ? prio : nullptr;
    a.active = greedy_mode == 2 ? active : nullptr;
    a.cid = greedy_mode == 2 ? cid : nullptr;
    a.beam = beam;
    big2_fill_model(a, m, x_const);
    size_t lds = std::max(b2_lds_bytes(a.max_nodes, &a.lds_cnt_off, &a.lds_perm_off, &a.lds_stage_off, &a.lds_tab_off),
                          b2_lgs_lds(a.max_nodes));
    a.by_priority = by_priority;
    if (a.cid && whole_step) {
        // the completions and the pick in this launch too (rollout_bits.h) when sixteen candidates do, every graph's columns fit
        // what the search has left of the Z1 space, and the instances' state words fit behind the selection's scratch.
        // option "rollout_bits" = 0: general.hip's launches.
        const bool bits_off = opt(OPT_ROLLOUT_BITS) == 0;
        const size_t cl_off = b2_lgs_lds(a.max_nodes) - 16;  // (where the search puts the 16-bit columns)
        const size_t cl_cap = (size_t)a.max_nodes * 64 > cl_off ? ((size_t)a.max_nodes * 64 - cl_off) / 2 : 0;
        const size_t roll = ((size_t)a.lds_stage_off + cand_scratch_bytes(kB2Block) + 15) & ~(size_t)15;
        const size_t need = roll + kCandMaxBeam * 4 + roll_lds_bytes(a.max_nodes);
        if (!bits_off && beam <= kRollBeam && weights && (size_t)std::max(b->max_graph_edges, 0) <= cl_cap && need <= 160 * 1024) {
            a.roll_off = (int32_t)roll;
            lds = std::max(lds, need);
            *whole_step = 1;
        }
    }
    return big2_launch(a, b->num_graphs, lds, "big_residual", s);
}

}  // namespace dgcn
