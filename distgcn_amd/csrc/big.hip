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
// The first layer on constant input features (the reference's row-normalised ones, gcn/utils.py:98-106) and the second
// layer's transform - the two chains the contract carries in double - run inside the launch too (values-only walk, f64 MFMA),
// and so does the last layer (32 -> 1: two fmaf chains round the row's four lanes, then a width-1 walk over the records).
// With explicit features the caller runs layer 0 and the transform of layer 1 with the layer-by-layer kernels first.
// Arithmetic as everywhere (include/dgcn.h): transform = k-ordered fmaf chain (v_mfma_f32_16x16x4_f32), aggregation = fmaf
// chain over the row's entries in storage order from 0, then Z0 + sum, + bias, activation: bit-identical to mode 0.
//
// Bound: the LDS array (one 128-byte row of Z1 per entry and layer: entries x 19 x 128 B / (128 B/clk) per graph);
// HBM sees the support once per launch (the records are re-read from L2 / MALL by every layer).
#include <algorithm>
#include <atomic>
#include <type_traits>

#include "common.h"
#include "lgs_rounds.h"
#include "big_common.h"
#include "cand_select.h"
#include "rollout_bits.h"

namespace dgcn {

// (DGCN_DIAG builds, option "diag_flags" bit 0: every walk reads its tile's FIRST record group again and again - wrong results, a timing
// experiment: what would the launch cost if the support's records did not have to stream from the L2 / MALL?)
#ifdef DGCN_DIAG
#define BIG_DIAG_SAME_GROUP (a.lgs_cols_lds == 2)
#else
#define BIG_DIAG_SAME_GROUP false
#endif
#ifdef DGCN_DIAG
#define BIG_STAMP(i)                                                   \
    do {                                                               \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();    \
        big_acc[i] += _t - big_t0;                                     \
        big_t0 = _t;                                                   \
    } while (0)
#define BIG_STAMP_FLUSH()                                                                                              \
    do {                                                                                                               \
        if (a.stamps && threadIdx.x == 0) {                                                                            \
            _Pragma("unroll") for (int _i = 0; _i < 12; ++_i) a.stamps[(size_t)g * 16 + _i] = big_acc[_i];           \
            a.stamps[(size_t)g * 16 + 13] = __builtin_amdgcn_s_memrealtime();                                          \
        }                                                                                                              \
    } while (0)
#else
#define BIG_STAMP(i) do { } while (0)
#define BIG_STAMP_FLUSH() do { } while (0)
#endif

// TILES: sixteen-row tiles a wave keeps in registers (4: up to 64 BLOCK / 64 tiles; 2: half of that, and the freed registers hold
// a second group of records in flight: the walk asks for its records TWO groups of four trips ahead)
// RESID: compiled with the residual-step code (big_residual); the plain kernels do not carry its state through their loops
template <int BLOCK, int TILES, bool RESID = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4))) void k_big(BigArgs a) {
    constexpr int kWavesB = BLOCK / 64;
    constexpr int DEPTH = TILES == 2 ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char big_lds[];
    const int g = blockIdx.x;
    const int n0 = a.graph_ptr[g], ng = a.graph_ptr[g + 1] - n0;
    if (RESID && a.cid && threadIdx.x < kCandMaxBeam) a.cid[(size_t)g * kCandMaxBeam + threadIdx.x] = -1;  // (a graph left alone has no candidates)
    if (ng <= 0) {
        if (a.do_lgs && threadIdx.x == 0) {  // an empty graph: no rounds, total 0 (heuristics.py's loops do not run)
            if (a.rounds) a.rounds[g] = 0;
            if (a.totals) a.totals[g] = 0.0;
        }
        return;
    }
    float* bufB = reinterpret_cast<float*>(big_lds);  // LDS offset 0: a gather address is the record's word ^ (chunk << 4)
    const unsigned zrow = (unsigned)a.max_nodes * 128u;
    unsigned short* cnt = reinterpret_cast<unsigned short*>(big_lds + a.lds_cnt_off);  // [max_nodes] entries per row (clamped: only orders rows and bounds walks)
    unsigned short* perm = reinterpret_cast<unsigned short*>(big_lds + a.lds_perm_off);  // [max_nodes] rows by descending entry count
    int* hist = reinterpret_cast<int*>(big_lds + a.lds_stage_off);                    // [kBigBins] (P0 only: the staging tiles' space)
    int* ttrips = reinterpret_cast<int*>(big_lds + a.lds_tab_off);                    // [64] trips per tile
    unsigned* tbase = reinterpret_cast<unsigned*>(ttrips + 64);                        // [64] first record of a tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2* rec = a.rec + (size_t)g * a.rec_cap;
    int fault = 0;
#ifdef DGCN_DIAG
    unsigned long long big_t0 = __builtin_amdgcn_s_memtime();
    unsigned long long big_acc[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)g * 16 + 12] = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- (residual step) which vertices are undecided, is anything left to do: the alive bytes sit in the Z1 space until the
    // first transform writes there
    constexpr bool resid = RESID;
    uint8_t* alive = reinterpret_cast<uint8_t*>(big_lds);
    int nr = ng;  // rows of the (residual) graph: its undecided vertices
    if (resid) {
        if (threadIdx.x < 2) ttrips[threadIdx.x] = 0;
        __syncthreads();
        int c_al = 0, pos = 0;
        for (int v = threadIdx.x; v < ng; v += BLOCK) {
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
        __syncthreads();  // (the tile table's space is written again below)
        // nothing left, or no positive weight left (np.sum(wts_nn) <= 0 -> break, mwis_gdpg_call.py:286): the graph is left alone
        if (!any_pos) {
            if (threadIdx.x == 0) {
                if (a.rounds) a.rounds[g] = 0;
                if (a.totals) a.totals[g] = 0.0;
                if (a.active) a.active[g] = 0;
            }
            for (int v = threadIdx.x; v < ng; v += BLOCK) a.scores[n0 + v] = 0.f;
            return;
        }
    }
    const int tiles = (nr + 15) >> 4;
    // ---- P0: row lengths, row order (counting sort, descending), Z1 of the first aggregation into LDS
    // One bin per possible entry count (a row of a 976-vertex graph has at most 977 entries; a caller's matrix with repeated
    // columns may exceed that: such rows share the last bin, and a tile's trips are the maximum over its sixteen rows, so no
    // row is ever cut short whatever order the bins give).
    for (int i = threadIdx.x; i < kBigBins; i += BLOCK) hist[i] = 0;
    if (threadIdx.x < 32) reinterpret_cast<float*>(big_lds + zrow)[threadIdx.x] = 0.f;
    __syncthreads();
    for (int v = threadIdx.x; v < ng; v += BLOCK) {
        unsigned c = a.arow ? (unsigned)(a.arow[n0 + v + 1] - a.arow[n0 + v]) + 1u : (unsigned)(a.lrow[n0 + v + 1] - a.lrow[n0 + v]);
        if (resid) {
            // degree in the residual graph; a decided vertex has no row at all (count 0: it sorts behind every undecided one)
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
        atomicAdd(&hist[min((int)c, kBigBins - 1)], 1);
    }
    for (int idx = threadIdx.x; idx < (a.front ? 0 : ng * 8); idx += BLOCK) {
        const int v = idx >> 3, c = idx & 7;
        const float4 z = *reinterpret_cast<const float4*>(a.Zin + (size_t)(n0 + v) * 64 + kBH + 4 * c);
        *reinterpret_cast<float4*>(bufB + v * kBH + ((c ^ big_key(v)) << 2)) = z;
    }
    __syncthreads();
    // start offset of count class c in the descending order = rows with a larger count: a suffix scan over the bins, PER
    // consecutive bins per thread (shuffle scan inside a wave, the waves' totals through the tile table's space)
    {
        constexpr int PER = kBigBins / BLOCK;
        static_assert(PER >= 1 && PER * BLOCK == kBigBins, "k_big: the bins are dealt evenly");
        int h[PER], own = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { h[j] = hist[threadIdx.x * PER + j]; own += h[j]; }
        int suf = own;  // inclusive suffix sum over the lanes of the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_down(suf, off);
            if (lane + off < 64) suf += t;
        }
        if (lane == 0) ttrips[wave] = suf;
        __syncthreads();
        int above = suf - own;
        for (int w = wave + 1; w < kWavesB; ++w) above += ttrips[w];
        __syncthreads();  // (the tile table's space is written again below)
#pragma unroll
        for (int j = PER - 1; j >= 0; --j) { hist[threadIdx.x * PER + j] = above; above += h[j]; }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < ng; v += BLOCK) {
        const int pos = atomicAdd(&hist[min((int)cnt[v], kBigBins - 1)], 1);
        perm[pos] = (unsigned short)v;  // (order among equal counts: whatever the atomics took - it decides which rows share a pass, never a sum)
    }
    __syncthreads();
    // trips of every tile (its longest row's) and where its records start: one wave, one scan
    if (wave == 0) {
        int longest = 0;
        if (lane < tiles) {
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int sl = lane * 16 + j;
                if (sl < nr) longest = max(longest, (int)cnt[perm[sl]]);
            }
        }
        const int tl = lane < tiles ? max(1, (longest + 3) >> 2) : 0;
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
    // (adjacency input) every vertex's d^-1/2 once, in the staging tiles' space (free until the first layer's epilogue): the
    // records below then need no global lookups beyond the column ids themselves
    double* dvl = reinterpret_cast<double*>(big_lds + a.lds_stage_off);
    if (a.arow) {
        for (int v = threadIdx.x; v < ng; v += BLOCK) {
            const int d = (int)cnt[v] - 1;
            double x = 0.0;
            if (d >= a.table_len) fault |= DGCN_FAULT_DEGREE_RANGE;
            else if (d >= 0) x = a.dinv[d];  // (a decided vertex of a residual step has no row: nobody reads its slot)
            dvl[v] = x;
        }
    }
    __syncthreads();
    BIG_STAMP(0);  // row lengths, row order, tiles, d^-1/2
    const int s16 = lane >> 2, kq4 = lane & 3;  // aggregation: row slot of the tile, quarter of the row
    // every wave writes the records of its own tiles (read back by the same lanes: no barrier)
    for (int t = wave; t < tiles; t += kWavesB) {
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
            // The row of the RESIDUAL graph: the diagonal, then the undecided neighbours in CSR order - the re-sliced row of
            // mwis_gdpg_call.py:284-285 without re-slicing.  The row's four lanes take four raw entries at a time, keep the undecided
            // ones and place them by a prefix count within the quad (entry p of the row is record 64 (p >> 2) + 4 s16 + (p & 3) of
            // the tile: written by whichever lane finds it, read by lane p & 3 of the quad after the workgroup barrier below).
            uint2* rowout = rec + base + s16 * 4;
            const int raw = has ? c - 1 : 0;
            if (has && kq4 == 0) rowout[0] = make_uint2(__float_as_uint(1.0f), big_word(v));
            int pos = has ? 1 : 0;
            for (int j0 = 0; __any(j0 < raw); j0 += 4) {  // (every lane of the wave takes part in the ballot)
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
                    if (p < 4 * trips) rowout[(p >> 2) * 64 + (p & 3)] = make_uint2(__float_as_uint((float)(-(dvl[u] * dv))), big_word(u));
                }
                pos += __popc(q);
            }
            for (int p = pos + kq4; p < 4 * trips; p += 4) rowout[(p >> 2) * 64 + (p & 3)] = nothing;
        } else
        for (int t0 = 0; t0 < trips; t0 += 4) {  // four trips' loads in flight (the walk is a chain of global round trips otherwise)
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
                    if (a.arow && e == 0) r = make_uint2(__float_as_uint(1.0f), big_word(v));  // (I - A_hat)[v][v], zero-diagonal adjacency
                    else {
                        const int u = uu[i];
                        if (u < 0 || u >= ng) fault |= DGCN_FAULT_BAD_COLUMN;
                        else if (a.arow) {
                            if (u == v) fault |= DGCN_FAULT_SELF_LOOP;
                            // reference order: (A_vu * dinv[u]) * dinv[v] in float64, negated by "eye - A_hat", then the float32 feed cast
                            r = make_uint2(__float_as_uint((float)(-(dvl[u] * dv))), big_word(u));
                        } else r = make_uint2(__float_as_uint(vv[i]), big_word(u));
                    }
                }
                out[(t0 + i) * 64] = r;
            }
        }
    }
    __syncthreads();  // (the d^-1/2 array shares the staging tiles' space)
    BIG_STAMP(1);  // records
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
    float pz[TILES][8];
    const unsigned voff = (unsigned)lane * 8u;
    BigRec4 A = {}, Bq = {}, Cq = {};  // the group being walked, the next one(s) in flight
    int pf_k = 0, pf_j = 0;
    (void)Cq;
#define DGCN_BQB(x, e) __builtin_amdgcn_update_dpp(0, (int)(x), (e) * 0x55, 0xf, 0xf, true)
#define DGCN_BQBF(x, e) __int_as_float(DGCN_BQB(__float_as_int(x), e))
    // the tile of slot k of this wave: t, its trips and records, the wave's next tile's records
#define DGCN_BTILE_HEAD                                                                                                  \
        const int t = wave + kWavesB * k;                                                                              \
        const int trips = __builtin_amdgcn_readfirstlane(ttrips[t]);                                                   \
        const int slot = t * 16 + s16;                                                                                 \
        const bool has = slot < nr;                                                                                    \
        (void)has;
    // The records of a wave's tiles are one stream of GROUPS (four trips each, tile after tile); every trip needs a record
    // from global memory (L2 / MALL: 500 .. 2 000 cycles), so the stream is requested DEPTH groups ahead of the walk - a
    // cursor (pf_k, pf_j) names the next group to ask for, whichever tile it belongs to.  (With one group in flight per wave
    // a CU has 32 KB of records under way: 3.3 TB/s chip-wide at MALL latency - what ER(500, 0.1) measured; TILES = 2 frees
    // the registers for a second one.)
#define DGCN_BPREFETCH(X)                                                                                                \
        {                                                                                                              \
            const int pt_ = wave + kWavesB * pf_k;                                                                     \
            if (pf_k < TILES && pt_ < tiles) { /* (wave-uniform) */                                                    \
                const int ptr_ = __builtin_amdgcn_readfirstlane(ttrips[pt_]);                                          \
                const unsigned pb_ = (unsigned)__builtin_amdgcn_readfirstlane((int)tbase[pt_]);                        \
                big_load_group(X, reinterpret_cast<const char*>(rec + pb_) + (size_t)(BIG_DIAG_SAME_GROUP ? 0 : pf_j) * 2048 + voff); /* (its last trips may lie past the tile: never walked) */ \
                pf_j += 1;                                                                                             \
                if (pf_j * 4 >= ptr_) { pf_j = 0; pf_k += 1; }                                                         \
            }                                                                                                          \
        }
#define DGCN_BWALK(TRIP)                                                                                                 \
        for (int g0 = 0; g0 < trips; g0 += 4) { /* (trips is wave-uniform: scalar branches) */                         \
            if constexpr (DEPTH == 2) { DGCN_BPREFETCH(Cq) } else { DGCN_BPREFETCH(Bq) }                               \
            TRIP(A.r0, g0)                                                                                             \
            if (g0 + 1 < trips) TRIP(A.r1, g0 + 1)                                                                     \
            if (g0 + 2 < trips) TRIP(A.r2, g0 + 2)                                                                     \
            if (g0 + 3 < trips) TRIP(A.r3, g0 + 3)                                                                     \
            A = Bq;                                                                                                    \
            if constexpr (DEPTH == 2) Bq = Cq;                                                                         \
        }
#define DGCN_BFIRST_GROUP                                                                                                \
        pf_k = 0;                                                                                                      \
        pf_j = 0;                                                                                                      \
        DGCN_BPREFETCH(A)                                                                                              \
        if constexpr (DEPTH == 2) { DGCN_BPREFETCH(Bq) }
    // aggregation layout (row s16, chunks cfirst / csecond: oA, oB) -> the MFMA's operand layout (lane 16 q + r: H[r][4 s + q])
#define DGCN_BSTAGE_TO_OPERAND(OA, OB)                                                                                   \
        *reinterpret_cast<float4*>(stg + s16 * kBH + ((cfirst ^ (s16 & 7)) << 2)) = OA;                                \
        *reinterpret_cast<float4*>(stg + s16 * kBH + ((csecond ^ (s16 & 7)) << 2)) = OB;                               \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */                                                           \
        __builtin_amdgcn_wave_barrier();                                                                               \
        _Pragma("unroll") for (int s = 0; s < 8; ++s) pz[k][s] = stg[mr * kBH + (((s ^ (mr & 7)) << 2) | mq)];         \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        __builtin_amdgcn_wave_barrier(); /* (the staging tile is rewritten by the next tile) */
    // MFMA output layout (row mr, chunks mq and 4 + mq: O0, O1) -> aggregation layout, into pz[k]
#define DGCN_BSTAGE_TO_AGG(O0, O1)                                                                                       \
        *reinterpret_cast<float4*>(stg + mr * kBH + ((mq ^ (mr & 7)) << 2)) = O0;                                      \
        *reinterpret_cast<float4*>(stg + mr * kBH + (((4 + mq) ^ (mr & 7)) << 2)) = O1;                                \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        __builtin_amdgcn_wave_barrier();                                                                               \
        {                                                                                                              \
            const float4 yA = *reinterpret_cast<const float4*>(stg + s16 * kBH + ((cfirst ^ (s16 & 7)) << 2));        \
            const float4 yB = *reinterpret_cast<const float4*>(stg + s16 * kBH + ((csecond ^ (s16 & 7)) << 2));       \
            pz[k][0] = yA.x; pz[k][1] = yA.y; pz[k][2] = yA.z; pz[k][3] = yA.w;                                        \
            pz[k][4] = yB.x; pz[k][5] = yB.y; pz[k][6] = yB.z; pz[k][7] = yB.w;                                        \
        }                                                                                                              \
        __builtin_amdgcn_s_waitcnt(0xC07F);                                                                            \
        __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < TILES; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) pz[k][j] = 0.f;

    if (a.front) {
        // -------- layer index 0 on constant input features (X == NULL: what every script of the reference feeds,
        // gcn/utils.py:98-106) and the transform of layer index 1, the two places whose chains run in double (include/dgcn.h).
        // Every row of Z = x.[W0 | W1] is the same 64 numbers, so the aggregation's chain fma(val_j, Z1[u_j][c], acc) needs
        // the entries' VALUES only - no gathers (fused.hip's const_rows).  The neutral records do not serve here (-0.0 times a
        // negative feature is +0.0, and -0.0 + +0.0 is +0.0): entries past the row's end are skipped by count.
        const BigFront& F = a.first;
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
        DGCN_BFIRST_GROUP
#pragma unroll
        for (int k = 0; k < TILES; ++k) {
            if (wave + kWavesB * k < tiles) {  // (wave-uniform)
                DGCN_BTILE_HEAD
                const int crow = has ? (int)cnt[perm[slot]] : 0;
                double accd[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) accd[j] = 0.0;
#define DGCN_BTRIP_FRONT(R, TT)                                                                                          \
                {                                                                                                      \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
                        const double ad = (double)__int_as_float(e == 0 ? DGCN_BQB(R.x, 0) : e == 1 ? DGCN_BQB(R.x, 1) : e == 2 ? DGCN_BQB(R.x, 2) : DGCN_BQB(R.x, 3)); \
                        if (4 * (TT) + e < crow) {                                                                     \
                            _Pragma("unroll") for (int j = 0; j < 8; ++j) accd[j] = fma(ad, z1d[j], accd[j]);          \
                        }                                                                                              \
                    }                                                                                                  \
                }
                DGCN_BWALK(DGCN_BTRIP_FRONT)
#undef DGCN_BTRIP_FRONT
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    double d = (double)z0c[j] + accd[j];
                    if (F.bias0) d += (double)F.bias0[(j < 4 ? 4 * cfirst : 4 * csecond) + (j & 3)];
                    o[j] = big_act((float)d, F.act0);
                }
                const float4 oA = make_float4(o[0], o[1], o[2], o[3]), oB = make_float4(o[4], o[5], o[6], o[7]);
                DGCN_BSTAGE_TO_OPERAND(oA, oB)
            }
        }
        // transform of layer index 1: every chain in double, rounded once (v_mfma_f64_16x16x4_f64 = the ascending fma chain;
        // its output register i of lane (q, r) holds row 4 i + q: the weight fragments are loaded accordingly, fused.hip)
        big_load_bfrag(F.W1, bfrag, true);
#pragma unroll
        for (int k = 0; k < TILES; ++k) {
            const int t = wave + kWavesB * k;
            if (t < tiles) {
                float4 zo[4];
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    bf64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const double ad = (double)pz[k][s];
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)bfrag[s][2 * cp], ad, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)bfrag[s][2 * cp + 1], ad, acc1, 0, 0, 0);
                    }
                    zo[2 * cp] = make_float4((float)acc0[0], (float)acc0[1], (float)acc0[2], (float)acc0[3]);
                    zo[2 * cp + 1] = make_float4((float)acc1[0], (float)acc1[1], (float)acc1[2], (float)acc1[3]);
                }
                const int mslot = t * 16 + mr;
                if (mslot < nr) {
                    const int v = (int)perm[mslot];
                    *reinterpret_cast<float4*>(bufB + v * kBH + ((mq ^ big_key(v)) << 2)) = zo[2];
                    *reinterpret_cast<float4*>(bufB + v * kBH + (((4 + mq) ^ big_key(v)) << 2)) = zo[3];
                }
                DGCN_BSTAGE_TO_AGG(zo[0], zo[1])
            }
        }
        __syncthreads();  // Z1 of layer index 1 is complete
    } else {
        // Z0 of the first aggregation: from the caller's Z (row-major, Z0 | Z1), in the aggregation's layout
#pragma unroll
        for (int k = 0; k < TILES; ++k) {
            const int slot = (wave + kWavesB * k) * 16 + s16;
            if (slot < ng) {
                const float* zr = a.Zin + (size_t)(n0 + (int)perm[slot]) * 64;
                const float4 yA = *reinterpret_cast<const float4*>(zr + 4 * cfirst), yB = *reinterpret_cast<const float4*>(zr + 4 * csecond);
                pz[k][0] = yA.x; pz[k][1] = yA.y; pz[k][2] = yA.z; pz[k][3] = yA.w;
                pz[k][4] = yB.x; pz[k][5] = yB.y; pz[k][6] = yB.z; pz[k][7] = yB.w;
            }
        }
    }
    float zz0[TILES], zz1[TILES];  // the last layer's (width 1) z0 / z1 of this lane's row
#pragma unroll
    for (int k = 0; k < TILES; ++k) zz0[k] = zz1[k] = 0.f;
    BIG_STAMP(2);  // layer 0 and the transform of layer 1 (or Z of layer 1 read from global memory)
    for (int i = 0; i < a.num_hidden; ++i) {
        const BigLayer& L = a.layers[i];
        const bool last = i == a.num_hidden - 1;
        // -------- aggregation: H' = act(Z0 + L.Z1 + b), 4 lanes x 2 float4 per row, 16 rows per pass
        DGCN_BFIRST_GROUP
#pragma unroll
        for (int k = 0; k < TILES; ++k) {
            if (wave + kWavesB * k < tiles) {  // (wave-uniform)
                DGCN_BTILE_HEAD
                float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = accA;
#define DGCN_BTRIP(R, TT)                                                                                                \
                {                                                                                                      \
                    float4 zA[4], zB[4];                                                                               \
                    float av[4];                                                                                       \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
                        const unsigned w = (unsigned)(e == 0 ? DGCN_BQB(R.y, 0) : e == 1 ? DGCN_BQB(R.y, 1) : e == 2 ? DGCN_BQB(R.y, 2) : DGCN_BQB(R.y, 3)); \
                        av[e] = __int_as_float(e == 0 ? DGCN_BQB(R.x, 0) : e == 1 ? DGCN_BQB(R.x, 1) : e == 2 ? DGCN_BQB(R.x, 2) : DGCN_BQB(R.x, 3)); \
                        zA[e] = big_lds_chunk(w ^ cA);                                                                 \
                        zB[e] = big_lds_chunk(w ^ cB);                                                                 \
                    }                                                                                                  \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
                        accA = big_fma4(av[e], zA[e], accA);                                                           \
                        accB = big_fma4(av[e], zB[e], accB);                                                           \
                    }                                                                                                  \
                }
                DGCN_BWALK(DGCN_BTRIP)
#undef DGCN_BTRIP
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
                    // the last layer (32 -> 1): z = H'.[w0 | w1], two fmaf chains over k = 0..31 - the row's features sit in its
                    // four lanes, chunk c in lane c & 3 (as its first or its second float4): the chains go round the quad
                    const float* Wl = a.Wlast;
                    float q0 = 0.f, q1 = 0.f;
#define DGCN_BLAST_STEP(C)                                                                                               \
                    {                                                                                                  \
                        const float4 o = (cfirst == (C)) ? oA : oB;                                                    \
                        float t0 = q0, t1 = q1;                                                                        \
                        t0 = fmaf(o.x, Wl[(4 * (C) + 0) * 2], t0); t1 = fmaf(o.x, Wl[(4 * (C) + 0) * 2 + 1], t1);      \
                        t0 = fmaf(o.y, Wl[(4 * (C) + 1) * 2], t0); t1 = fmaf(o.y, Wl[(4 * (C) + 1) * 2 + 1], t1);      \
                        t0 = fmaf(o.z, Wl[(4 * (C) + 2) * 2], t0); t1 = fmaf(o.z, Wl[(4 * (C) + 2) * 2 + 1], t1);      \
                        t0 = fmaf(o.w, Wl[(4 * (C) + 3) * 2], t0); t1 = fmaf(o.w, Wl[(4 * (C) + 3) * 2 + 1], t1);      \
                        q0 = DGCN_BQBF(t0, (C) & 3);                                                                   \
                        q1 = DGCN_BQBF(t1, (C) & 3);                                                                   \
                    }
                    DGCN_BLAST_STEP(0) DGCN_BLAST_STEP(1) DGCN_BLAST_STEP(2) DGCN_BLAST_STEP(3)
                    DGCN_BLAST_STEP(4) DGCN_BLAST_STEP(5) DGCN_BLAST_STEP(6) DGCN_BLAST_STEP(7)
#undef DGCN_BLAST_STEP
                    zz0[k] = q0;
                    zz1[k] = q1;
                } else {
                    DGCN_BSTAGE_TO_OPERAND(oA, oB)
                }
            }
        }
        BIG_STAMP(3);  // aggregations (sum over the hidden layers)
#ifdef DGCN_DIAG
        if (a.stamps && lane == 0) a.stamps[(size_t)a.num_graphs_diag * 16 + (size_t)g * 16 + wave] = big_acc[3];  // every wave's own sum, behind the phase table
#endif
        if (last) break;
        // the next layer's weight fragments: requested here, they land while this wave waits at the barrier
        big_load_bfrag(L.Wnext, bfrag, false);
        BIG_STAMP(4);  // issuing the weight loads
        __syncthreads();  // every gather of this layer has read Z1
        BIG_STAMP(5);  // barrier behind the aggregation
        // -------- transform of the next layer: Z0 | Z1 = H'.[W0 | W1], v_mfma_f32_16x16x4_f32, operands swapped (D^T = W^T.H^T)
        // so that a lane ends with four consecutive features of one vertex; Z1 -> bufB, Z0 -> registers (aggregation layout)
#pragma unroll
        for (int k = 0; k < TILES; ++k) {
            const int t = wave + kWavesB * k;
            if (t < tiles) {
                bf32x4 acc[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) acc[ct] = (bf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfrag[s][ct], pz[k][s], acc[ct], 0, 0, 0);
                const int mslot = t * 16 + mr;
                if (mslot < nr) {
                    const int v = (int)perm[mslot];
#pragma unroll
                    for (int ct = 2; ct < 4; ++ct) {
                        const int chunk = (ct & 1) * 4 + mq;
                        *reinterpret_cast<float4*>(bufB + v * kBH + ((chunk ^ big_key(v)) << 2)) = make_float4(acc[ct][0], acc[ct][1], acc[ct][2], acc[ct][3]);
                    }
                }
                const float4 o0 = make_float4(acc[0][0], acc[0][1], acc[0][2], acc[0][3]), o1 = make_float4(acc[1][0], acc[1][1], acc[1][2], acc[1][3]);
                DGCN_BSTAGE_TO_AGG(o0, o1)
            }
        }
        BIG_STAMP(6);  // transforms (sum)
        __syncthreads();  // Z1 of the next layer is complete
        BIG_STAMP(7);  // barrier behind the transform
    }
    // -------- the last layer's aggregation at width 1: score = act(z0 + sum_j val_j z1[u_j] + b), an fmaf chain in storage
    // order like every other; z1 of the whole graph as a float array over bufB (the record's word >> 7 is the neighbour)
    __syncthreads();  // every gather of the last hidden aggregation has read Z1
    float* zl = bufB;
#pragma unroll
    for (int k = 0; k < TILES; ++k) {
        const int slot = (wave + kWavesB * k) * 16 + s16;
        if (slot < nr && kq4 == 0) zl[perm[slot]] = zz1[k];
    }
    if (threadIdx.x == 0) zl[a.max_nodes] = 0.f;  // the neutral record's neighbour
    __syncthreads();
    DGCN_BFIRST_GROUP
#pragma unroll
    for (int k = 0; k < TILES; ++k) {
        if (wave + kWavesB * k < tiles) {
            DGCN_BTILE_HEAD
            float accs = 0.f;
#define DGCN_BTRIP_TAIL(R, TT)                                                                                           \
            {                                                                                                          \
                float zs[4], av[4];                                                                                    \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                        \
                    const unsigned w = (unsigned)(e == 0 ? DGCN_BQB(R.y, 0) : e == 1 ? DGCN_BQB(R.y, 1) : e == 2 ? DGCN_BQB(R.y, 2) : DGCN_BQB(R.y, 3)); \
                    av[e] = __int_as_float(e == 0 ? DGCN_BQB(R.x, 0) : e == 1 ? DGCN_BQB(R.x, 1) : e == 2 ? DGCN_BQB(R.x, 2) : DGCN_BQB(R.x, 3)); \
                    zs[e] = zl[w >> 7];                                                                                \
                }                                                                                                      \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) accs = fmaf(av[e], zs[e], accs);                         \
            }
            DGCN_BWALK(DGCN_BTRIP_TAIL)
#undef DGCN_BTRIP_TAIL
            float o = zz0[k] + accs;
            if (a.bias_last) o += a.bias_last[0];
            o = big_act(o, a.act_last);
            if (has && kq4 == 0 && trips > 0) {
                const int v = (int)perm[slot];
                a.scores[n0 + v] = o;
                if (a.do_lgs) {  // the priority (mwis_dqn_call.py:232: float32 x float64 -> float64), behind z1's array in LDS
                    double p = (double)o;
                    if (a.predict_mwis && a.weights) p *= a.weights[n0 + v];
                    reinterpret_cast<double*>(big_lds + big_lgs_base(a.max_nodes))[v] = p;
                }
            }
        }
    }
    BIG_STAMP(8);  // last layer: width-1 walk, scores, priorities
    if (a.do_lgs) {
        // -------- the local greedy search (heuristics.py:77-116), as k_lgs runs it: priorities, state bytes, row bounds and the
        // graph's 16-bit local column ids in LDS (everything the forward pass kept there is dead after the next barrier)
        double* pr = reinterpret_cast<double*>(big_lds + big_lgs_base(a.max_nodes));
        double* red = pr + a.max_nodes;
        unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(red + 1024);
        // The search on bit masks (lgs_rounds.h: lgs_mask_build / lgs_mask_rounds; a workgroup has a thread per vertex here,
        // ng <= BLOCK): ER(500, 0.1) 412 -> 397 us per launch against the rounds that walked 16-bit column lists in LDS.
        unsigned long long* liveA = acc64 + 4;   // [16] live vertices, a word per wave (ping)
        unsigned long long* liveB = liveA + 16;  // [16] (pong)
        unsigned long long* wonm = liveB + 16;   // [16] this round's winners
        uint8_t* st = reinterpret_cast<uint8_t*>(wonm + 16);
        unsigned long long* am = reinterpret_cast<unsigned long long*>(st + ((a.max_nodes + 15) & ~15));  // [ng][W64] ahead masks
        __syncthreads();  // every score of the graph is computed, every walk over z1 done
        const int W64 = (ng + 63) >> 6;
        const int tv = threadIdx.x;
        // (residual step) who takes part: the vertices that were undecided when the launch began (read again: the Z1 space held the
        // alive bytes only until the first transform).  A decided vertex reports score 0 and carries a NaN priority - never ahead
        // of anybody in the masks below.
        bool part = tv < ng;
        if (resid && tv < ng) {
            part = a.state[n0 + tv] == 0;
            if (!part) {
                a.scores[n0 + tv] = 0.f;
                pr[tv] = __longlong_as_double(0x7ff8000000000001ll);
            }
        }
        int bad = 0;
        if (tv < ng) {
            const double p = pr[tv];
            bad = part && p != p;
            st[tv] = 0;
        }
        // (no __syncthreads_or: ockl's workgroup reductions bring static LDS with them, and bufB must stay at LDS offset 0)
        if (threadIdx.x == 0) acc64[3] = 0;
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
            if (!resid) for (int v = threadIdx.x; v < ng; v += BLOCK) a.state[n0 + v] = 0;
            return;
        }
        if (resid && threadIdx.x == 0) {
            if (a.progress) atomicAdd(a.progress, 1);
            if (a.tail_word) atomicMax(a.tail_word, a.tail_tag | (unsigned long long)(unsigned)nr);
        }
        if (resid && a.greedy_mode == 2) {
            // the rollout: priorities out (a decided vertex: 0, as k_res_scatter leaves it); candidates, instances, completions
            // and the pick run right here (rollout_bits.h) or, beyond sixteen candidates, as general.hip's launches
            if (tv < ng) a.prio_out[n0 + tv] = part ? pr[tv] : 0.0;
            if (threadIdx.x == 0 && a.active) a.active[g] = 1;
            if (fault) atomicOr(a.status, fault);
            if (a.cid) {  // the candidates right here (a thread per vertex, the priorities at hand): no k_res_cand launch
                int32_t* cid = a.cid + (size_t)g * kCandMaxBeam;
                double pv[kCandPer];
#pragma unroll
                for (int i = 0; i < kCandPer; ++i) pv[i] = 0.0;
                pv[0] = part ? pr[tv] : 0.0;
                if (a.roll_off) {
                    // the whole step here: candidates (list mirrored in LDS), the completions of all of them at once, the pick.
                    // Behind the search's state bytes: row bounds | candidates | instance words | selection scratch | 16-bit columns
                    unsigned char* R = big_lds + a.roll_off;
                    int* rol = reinterpret_cast<int*>(R);
                    int32_t* cidl = rol + ((a.max_nodes + 1 + 3) & ~3);
                    unsigned char* extra = reinterpret_cast<unsigned char*>(cidl + kCandMaxBeam);
                    unsigned char* scratch = extra + roll_lds_bytes(a.max_nodes);
                    uint16_t* cl = reinterpret_cast<uint16_t*>(scratch + ((cand_scratch_bytes(BLOCK) + 15) & ~(size_t)15));
                    if (tv < ng) st[tv] = part ? 0 : 3;
                    if (threadIdx.x < kCandMaxBeam) cidl[threadIdx.x] = -1;
                    __syncthreads();
                    cand_select<BLOCK>(pv, part ? 1u : 0u, 1, min(a.beam, kCandMaxBeam), cid, scratch, cidl);
                    if (!a.by_priority && part) pr[tv] = a.weights[n0 + tv];  // (the completions go by weight: mwis_gdpg_call.py:640)
                    for (int v = threadIdx.x; v <= ng; v += BLOCK) rol[v] = a.arow[n0 + v];
                    const int e0 = a.arow[n0], e1 = a.arow[n0 + ng];
                    for (int base = e0 + threadIdx.x; base < e1; base += BLOCK * 4) {  // four loads in flight per thread
                        int c[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) c[i] = (base + i * BLOCK < e1) ? a.acol[base + i * BLOCK] : n0;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (base + i * BLOCK < e1) {
                                const int u = c[i] - n0;
                                cl[base + i * BLOCK - e0] = (uint16_t)((unsigned)u < (unsigned)ng ? u : ng);
                            }
                    }
                    __syncthreads();
                    RollArgs r;
                    r.ng = ng; r.n0 = n0; r.e0 = e0;
                    r.key = pr; r.st = st; r.rol = rol; r.cl = cl; r.cidl = cidl; r.beam = a.beam;
                    r.extra = extra;
                    r.max_nodes = a.max_nodes;
                    r.wl = a.by_priority ? nullptr : pr; r.weights = a.weights; r.state = a.state; r.rounds = a.rounds; r.totals = a.totals;
                    rollout_bits<BLOCK>(r, g);
                    return;
                }
                __syncthreads();  // (the staging tiles' space is free: the last walk is behind the barriers above)
                cand_select<BLOCK>(pv, part ? 1u : 0u, 1, min(a.beam, kCandMaxBeam), cid, big_lds + a.lds_stage_off);
            }
            return;
        }
        if (resid && a.greedy_mode == 1) {
            // solve_mwis_cit: the best-priority undecided vertex joins (np.argmax: lowest index among equals), its neighbours leave
            double bp = part ? pr[tv] : 0.0;
            int bv = part ? tv : -1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double op = __shfl_xor(bp, off);
                const int ov = __shfl_xor(bv, off);
                if (ov >= 0 && (bv < 0 || op > bp || (op == bp && ov < bv))) { bp = op; bv = ov; }
            }
            int* ri = reinterpret_cast<int*>(red + 16);
            if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = bp; ri[threadIdx.x >> 6] = bv; }
            __syncthreads();
            bp = red[0]; bv = ri[0];
#pragma unroll
            for (int w = 1; w < BLOCK / 64; ++w) {
                const double op = red[w];
                const int ov = ri[w];
                if (ov >= 0 && (bv < 0 || op > bp || (op == bp && ov < bv))) { bp = op; bv = ov; }
            }
            if (bv >= 0) {
                for (int j = a.arow[n0 + bv] + (int)threadIdx.x; j < a.arow[n0 + bv + 1]; j += BLOCK) {
                    const int u = a.acol[j] - n0;
                    if ((unsigned)u < (unsigned)ng && u != bv && a.state[n0 + u] == 0) a.state[n0 + u] = 2;
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
        lgs_mask_build<BLOCK>(a.arow, a.acol, n0, ng, pr, am, W64);
        const int rounds = lgs_mask_rounds<BLOCK>(tv, part, am, W64, liveA, liveB, wonm, st, resid ? a.max_rounds : 0);
        if (threadIdx.x == 0 && a.rounds) a.rounds[g] = rounds;
        {
            // state out; total weight of the set: the reduction tree of k_lgs<.., 1024> (strided partials, folded to 256 slots).
            // A residual step writes what it decided and counts what joined in THIS step; whoever was decided before stays as it was.
            double partial = 0.0;
            for (int v = threadIdx.x; v < ng; v += BLOCK) {
                const uint8_t s1 = st[v];
                if (resid) {
                    if (s1 != 0) {  // (only vertices that took part get a state in the rounds)
                        if (a.totals && s1 == 1) partial += a.weights ? a.weights[n0 + v] : pr[v];
                        a.state[n0 + v] = s1;
                    }
                } else {
                    if (a.totals && s1 == 1) partial += a.weights ? a.weights[n0 + v] : pr[v];
                    a.state[n0 + v] = s1;
                }
            }
            red[threadIdx.x] = partial;
        }
        if (a.totals) {
            __syncthreads();
            if (threadIdx.x < 256) {
                double acc = red[threadIdx.x];
                for (int k2 = 256; k2 < BLOCK; k2 += 256) acc += red[threadIdx.x + k2];
                red[threadIdx.x] = acc;
            }
            __syncthreads();
            for (int off = 128; off > 0; off >>= 1) {
                if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
                __syncthreads();
            }
            if (threadIdx.x == 0) a.totals[g] = red[0];
        }
    }
    BIG_STAMP(9);  // the search, state, totals
    BIG_STAMP_FLUSH();
#undef DGCN_BSTAGE_TO_AGG
#undef DGCN_BSTAGE_TO_OPERAND
#undef DGCN_BFIRST_GROUP
#undef DGCN_BPREFETCH
#undef DGCN_BWALK
#undef DGCN_BTILE_HEAD
#undef DGCN_BQBF
#undef DGCN_BQB
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

// tuning / A-B switches of this file (dgcn_set_option, options.h): "big" = 0 (no k_big: layer by layer), "big_solve" = 0
// (supports and greedy search in launches of their own), "big_block" = 512 | 1024, "big_tiles" = 4
static int big_env_enabled() { return opt(OPT_BIG); }
static int big_env_solve() { return opt(OPT_BIG_SOLVE); }
static int big_env_block() { return opt(OPT_BIG_BLOCK); }
static int big_env_tiles() { return opt(OPT_BIG_TILES); }

// threads per workgroup: a wave keeps at most kBigTilesPerWave tiles in registers, so 512 threads reach 512 vertices - and two
// such workgroups share a CU (the recipe of fused.hip's C3 launch: one's barriers and round trips under the other's work);
// larger graphs get the CU to themselves.  option "big_block" = 512 | 1024 overrides (tuning / tests).
static int big_block(int max_nodes) {
    int block = max_nodes <= 16 * kBigTilesPerWave * 8 ? 512 : 1024;
    const int want = big_env_block();
    if (want == 1024 || (want == 512 && max_nodes <= 16 * kBigTilesPerWave * 8)) block = want;
    return block;
}

static int big_rec_cap(const DgcnBatch* b) {
    // fused.hip's bound for block-major records of rows in descending order: entries + 20 N + 192 (+ one cache line)
    return ((b->max_graph_edges + b->max_nodes + 2 + 16 + 15) & ~15) + ((20 * b->max_nodes + 448 + 15) & ~15);
}

static size_t big_lds_bytes(int max_nodes, int block, int* cnt_off, int* perm_off, int* stage_off, int* tab_off) {
    size_t off = (size_t)max_nodes * 128 + 128;  // Z1 + the zero row
    *stage_off = (int)off;
    off += (size_t)std::max(block / 64 * 2048, 8192);  // a 16 x 32 float tile per wave (P0: the count histogram, the d^-1/2 array)
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
    if (big_env_enabled() == 0) return 0;
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

static int big_launch(BigArgs& a, int B, size_t lds, int block, const char* family, hipStream_t s);
static int big_block(int max_nodes);

// The forward pass in one launch (constant input features: X == NULL), or - explicit features - layer 0 and the transform
// of layer 1 by the layer-by-layer kernels exactly as layered_forward runs them, then everything else in one launch.
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
    const DgcnLayer& LL = m->layers_host[Lc - 1];
    int rc = DGCN_OK;
    const bool front = X == nullptr && L0.in_dim <= 64;  // constant input features: layer 0 and the transform of layer 1 inside the launch
    if (!front) {
        rc = transform_dispatch(X, L0.in_dim, x_const, b->num_nodes, L0.in_dim, L0.weights, 2 * kBH, Zbuf, 2 * kBH, s);
        if (rc) return rc;
        rc = spmm_f64acc_dispatch(lap, b->graph_ptr, b->num_graphs, b->max_nodes, Zbuf + kBH, 2 * kBH, kBH, Zbuf, 2 * kBH, L0.bias, L0.act,
                                  Hbuf, kBH, s);
        if (rc) return rc;
        rc = transform_f64acc_dispatch(Hbuf, kBH, x_const, b->num_nodes, kBH, L1.weights, 2 * kBH, Zbuf, 2 * kBH, s);
        if (rc) return rc;
    }
    BigArgs a = {};
    a.graph_ptr = b->graph_ptr;
    a.lrow = lap->row_ptr; a.lcol = lap->col_idx; a.lval = lap->values;
    a.Zin = Zbuf; a.rec = rec; a.status = status;
    a.front = front ? 1 : 0;
    a.first.W0 = L0.weights; a.first.bias0 = L0.bias; a.first.W1 = L1.weights; a.first.x_const = x_const;
    a.first.cin = L0.in_dim; a.first.act0 = L0.act;
    a.Wlast = LL.weights; a.bias_last = LL.bias; a.act_last = LL.act; a.scores = scores;
    a.rec_cap = big_rec_cap(b);
    a.max_nodes = (std::max(b->max_nodes, 16) + 15) & ~15;
    a.num_hidden = Lc - 2;
    for (int l = 1; l <= Lc - 2; ++l) {
        const DgcnLayer& L = m->layers_host[l];
        a.layers[l - 1].bias = L.bias;
        a.layers[l - 1].act = L.act;
        a.layers[l - 1].Wnext = l < Lc - 2 ? m->layers_host[l + 1].weights : nullptr;
    }
    int block = big_block(a.max_nodes);
    size_t lds = big_lds_bytes(a.max_nodes, block, &a.lds_cnt_off, &a.lds_perm_off, &a.lds_stage_off, &a.lds_tab_off);
    if (block == 512 && lds > 80 * 1024 && big_env_block() < 0) {
        block = 1024;
        lds = big_lds_bytes(a.max_nodes, block, &a.lds_cnt_off, &a.lds_perm_off, &a.lds_stage_off, &a.lds_tab_off);
    }
    return big_launch(a, b->num_graphs, lds, block, "big_forward", s);
}

// bytes of LDS the search at the end of the launch needs (k_lgs's arrays behind z1's), or 0 when a graph's columns do not fit
static size_t big_lgs_lds(int max_nodes, int max_graph_edges) {
    (void)max_graph_edges;  // (the search works on a mask per vertex - a bit per vertex of the graph -, not on the column lists)
    const size_t pad = (size_t)((max_nodes + 15) & ~15);
    const size_t w64 = (size_t)(max_nodes + 63) / 64;
    const size_t need = big_lgs_base(max_nodes) + (size_t)max_nodes * 8 + 1024 * 8 + (4 + 3 * 16) * 8 + pad + (size_t)max_nodes * w64 * 8 + 16;
    return need <= 160 * 1024 ? need : 0;
}

// 1 = dgcn_solve_batch's whole path in ONE launch: adjacency in, set out (constant input features, k_big's shapes, the graphs'
// column ids fit the LDS next to the search's state)
int big_solve_takes(const DgcnBatch* b, const DgcnModel* m, const float* X) {
    if (big_env_solve() == 0) return 0;
    return !X && big_takes(b, m) && m->layers_host[0].in_dim <= 64 && big_lgs_lds((std::max(b->max_nodes, 16) + 15) & ~15, b->max_graph_edges) != 0;
}

static void big_fill_model(BigArgs& a, const DgcnModel* m, float x_const) {
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

template <int BLOCK, int TILES, bool RESID = false>
static int big_launch_b(BigArgs& a, int B, size_t lds, const char* family, hipStream_t s) {
    if (lds > 64 * 1024) {
        static std::atomic<int> reserved[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!reserved[dev & 63].load(std::memory_order_relaxed)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_big<BLOCK, TILES, RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return fail(DGCN_ERR_LAUNCH, "k_big: cannot reserve %zu bytes of LDS", lds);
            reserved[dev & 63].store(1, std::memory_order_relaxed);
        }
    }
    TimedLaunch t(family, s);
    DGCN_LAUNCH(t, (k_big<BLOCK, TILES, RESID>), dim3((unsigned)B), dim3(BLOCK), lds, s, a);
    return check_launch("k_big");
}

static int big_launch(BigArgs& a, int B, size_t lds, int block, const char* family, hipStream_t s) {
    if (lds > 160 * 1024) return fail(DGCN_ERR_UNSUPPORTED, "k_big: %zu bytes of LDS", lds);
#ifdef DGCN_DIAG
    a.stamps = reinterpret_cast<unsigned long long*>(static_cast<uintptr_t>(opt64(OPT_DIAG_STAMPS)));
    a.num_graphs_diag = B;
    if (opt(OPT_DIAG_FLAGS) & 1) a.lgs_cols_lds = 2;
#endif
    // two tiles per wave where that covers the largest graph: the freed registers keep a second group of records in flight
    const bool two = a.max_nodes <= 16 * 2 * (block / 64) && big_env_tiles() != 4;
    if (a.residual) {
        if (block == 512) return two ? big_launch_b<512, 2, true>(a, B, lds, family, s) : big_launch_b<512, 4, true>(a, B, lds, family, s);
        return two ? big_launch_b<1024, 2, true>(a, B, lds, family, s) : big_launch_b<1024, 4, true>(a, B, lds, family, s);
    }
    if (block == 512) return two ? big_launch_b<512, 2>(a, B, lds, family, s) : big_launch_b<512, 4>(a, B, lds, family, s);
    return two ? big_launch_b<1024, 2>(a, B, lds, family, s) : big_launch_b<1024, 4>(a, B, lds, family, s);
}

// A1-A10 in one launch for graphs beyond fused.hip's LDS budget: support construction while the records are written, every
// layer, priority, local greedy search.  `scores` must be given (dgcn_solve_batch passes scratch when the caller wants none).
int big_solve(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, float x_const, const double* weights,
              int32_t predict_mwis, float* scores, uint8_t* state, int32_t* rounds, double* totals, int32_t* status, void* bws,
              hipStream_t s) {
    BigArgs a = {};
    a.graph_ptr = b->graph_ptr;
    a.arow = b->row_ptr; a.acol = b->col_idx; a.dinv = dinv_table; a.table_len = table_len;
    a.rec = reinterpret_cast<uint2*>((reinterpret_cast<uintptr_t>(bws) + 255) & ~(uintptr_t)255);
    a.status = status;
    a.rec_cap = big_rec_cap(b);
    a.max_nodes = (std::max(b->max_nodes, 16) + 15) & ~15;
    a.front = 1;
    a.scores = scores;
    a.do_lgs = 1; a.predict_mwis = predict_mwis; a.lgs_cols_lds = 1;
    a.weights = weights; a.state = state; a.rounds = rounds; a.totals = totals;
    big_fill_model(a, m, x_const);
    int block = big_block(a.max_nodes);
    size_t lds = std::max(big_lds_bytes(a.max_nodes, block, &a.lds_cnt_off, &a.lds_perm_off, &a.lds_stage_off, &a.lds_tab_off),
                          big_lgs_lds(a.max_nodes, b->max_graph_edges));
    if (block == 512 && lds > 80 * 1024 && big_env_block() < 0) {  // no second workgroup on the CU anyway: all 16 waves for this graph
        block = 1024;
        lds = std::max(big_lds_bytes(a.max_nodes, block, &a.lds_cnt_off, &a.lds_perm_off, &a.lds_stage_off, &a.lds_tab_off),
                       big_lgs_lds(a.max_nodes, b->max_graph_edges));
    }
    return big_launch(a, b->num_graphs, lds, block, "big_solve", s);
}

// One step of dgcn_solve_residual_batch in ONE launch (constant input features, k_big's shapes): the residual graph's support
// from the adjacency and the running state, every layer, priorities, and the greedy step - local rounds (solve_mwis_dit,
// mwis_gdpg_call.py:278-318) or the central pick (solve_mwis_cit, :343-384); for the rollouts (:596-659) the priorities are
// left in `prio` and the candidates, all completions and the pick run at the end of the same launch (cand_select.h, rollout_bits.h; beyond sixteen
// candidates general.hip's k_lgs / k_res_pick launches follow instead).  option "big_residual" = 0: the compaction
// launches + k_big + k_lgs instead (tests compare the two).
int big_residual_takes(const DgcnBatch* b, const DgcnModel* m, const float* X, int32_t feature_mode, int32_t options) {
    const int on = opt(OPT_BIG_RESIDUAL);
    if (on == 0 || feature_mode != 0 || (options & DGCN_RESIDUAL_SCORES_GIVEN)) return 0;
    return big_solve_takes(b, m, X);
}

int big_residual(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, float x_const, const double* weights,
                 int32_t predict_mwis, int32_t greedy_mode, int32_t max_rounds, float* scores, uint8_t* state, int32_t* rounds,
                 double* totals, int32_t* progress, int32_t* status, double* prio, int32_t* active, int32_t* cid, int32_t beam,
                 int32_t by_priority, int32_t* whole_step, unsigned long long* tail_word, unsigned long long tail_tag, void* bws,
                 hipStream_t s) {
    if (whole_step) *whole_step = 0;
    BigArgs a = {};
    a.graph_ptr = b->graph_ptr;
    a.arow = b->row_ptr; a.acol = b->col_idx; a.dinv = dinv_table; a.table_len = table_len;
    a.rec = reinterpret_cast<uint2*>((reinterpret_cast<uintptr_t>(bws) + 255) & ~(uintptr_t)255);
    a.status = status;
    a.rec_cap = big_rec_cap(b);
    a.max_nodes = (std::max(b->max_nodes, 16) + 15) & ~15;
    a.front = 1;
    a.scores = scores;
    a.do_lgs = 1; a.predict_mwis = predict_mwis; a.lgs_cols_lds = 1;
    a.weights = weights; a.state = state; a.rounds = rounds; a.totals = totals;
    a.residual = 1; a.greedy_mode = greedy_mode; a.max_rounds = max_rounds;
    a.progress = progress; a.tail_word = tail_word; a.tail_tag = tail_tag;
    a.prio_out = greedy_mode == 2 ? prio : nullptr;
    a.active = greedy_mode == 2 ? active : nullptr;
    a.cid = greedy_mode == 2 ? cid : nullptr;
    a.beam = beam;
    big_fill_model(a, m, x_const);
    int block = big_block(a.max_nodes);
    size_t lds = std::max(big_lds_bytes(a.max_nodes, block, &a.lds_cnt_off, &a.lds_perm_off, &a.lds_stage_off, &a.lds_tab_off),
                          big_lgs_lds(a.max_nodes, b->max_graph_edges));
    if (block == 512 && lds > 80 * 1024 && big_env_block() < 0) {
        block = 1024;
        lds = std::max(big_lds_bytes(a.max_nodes, block, &a.lds_cnt_off, &a.lds_perm_off, &a.lds_stage_off, &a.lds_tab_off),
                       big_lgs_lds(a.max_nodes, b->max_graph_edges));
    }
    a.by_priority = by_priority;
    if (a.cid && whole_step) {
        // the completions and the pick in this launch too (rollout_bits.h) when sixteen candidates do and row bounds, instance
        // words, the selection's scratch and the graph's 16-bit columns fit behind the search's state bytes.
        // option "rollout_bits" = 0: general.hip's launches.
        const bool bits_off = opt(OPT_ROLLOUT_BITS) == 0;
        const size_t pad = (size_t)((a.max_nodes + 15) & ~15);
        const size_t roll = (big_lgs_base(a.max_nodes) + (size_t)a.max_nodes * 8 + 1024 * 8 + (4 + 3 * 16) * 8 + pad + 15) & ~(size_t)15;
        const size_t need = roll + (size_t)((a.max_nodes + 1 + 3) & ~3) * 4 + kCandMaxBeam * 4 + roll_lds_bytes(a.max_nodes) +
                            ((cand_scratch_bytes(block) + 15) & ~(size_t)15) + (size_t)std::max(b->max_graph_edges, 0) * 2 + 16;
        if (!bits_off && beam <= kRollBeam && weights && need <= 160 * 1024) {
            a.roll_off = (int32_t)roll;
            lds = std::max(lds, need);
            *whole_step = 1;
        }
    }
    return big_launch(a, b->num_graphs, lds, block, "big_residual", s);
}

}  // namespace dgcn
