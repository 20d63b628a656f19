// The tail of an iterative search: every remaining step of a graph inside ONE launch, once its residual graph is small.
//
// A step of solve_mwis_dit / _cit / _rollout (mwis_gdpg_call.py:278-318, 343-384, 596-659) re-slices the graph to the
// undecided vertices, runs the GCN on it and decides a few vertices; dgcn_solve_residual_batch does one such step per call
// (fused.hip's residual-graph kernel, general.hip beyond one CU's LDS).  Measured (tools/solo_floor.py, profiles/
// r04_solo_floor.txt): with almost nothing left a step still costs ~57 us, 31 - 35 of them outside the layers - launch,
// image build from the global adjacency, about ten dependent global round trips, forty workgroup barriers - and a rollout
// search of N = 500 graphs spends its last ~40 of ~109 steps on fewer than 64 vertices.
//
// k_tail takes a graph over once at most kTailMax = 64 vertices are undecided and runs ALL its remaining steps: one
// 256-thread workgroup per graph, the residual adjacency read once (64-bit neighbour masks in registers + per-row lists in
// storage order in LDS), state / weights / priorities in LDS, nothing but the layer weights read from global memory between
// steps.  Per step: residual degrees -> L = I - D^-1/2 A D^-1/2 (diagonal first) -> the layer stack (tile_ops.h: the same
// MFMA transforms and chain orders as k_fused, hence the same bits) -> priorities -> the greedy step on ballots:
// a vertex wins a local-greedy round iff no live neighbour is ahead of it, i.e. (ahead_mask & live_mask) == 0 - one AND per
// round instead of a walk over the adjacency.  Rollout completions run one candidate per wave.
//
// Results equal the step-by-step path's: graphs are independent and a step is a function of the state alone
// (tests/test_gpu_tail.py: every solver, final states against the per-step kernels and the oracle).
#include "common.h"
#include "tile_ops.h"
#include "wave_reduce.h"

namespace dgcn {

constexpr int kTailMax = 64;    // undecided vertices a graph may have left (one per lane of a wave, one bit of a mask)
constexpr int kTailBlock = 256;  // four waves: a 16-row tile each in the transforms, four lanes per row in the aggregations
constexpr int kTailMaxLayers = 64;

struct TailLayer {
    const float* W;     // [cin][2 * cout]
    const float* bias;  // [cout] or null
    int32_t cin, cout, act, pad;
};

struct TailArgs {
    const int32_t* graph_ptr;
    const int32_t* row_ptr;
    const int32_t* col_idx;
    uint8_t* state;            // in / out: 0 undecided, 1 member, 2 excluded
    const double* weights;     // or null
    const double* dinv_table;
    int32_t table_len;
    int32_t feature_mode;
    float x_const;
    int32_t predict_mwis, greedy_mode, max_rounds, beam, by_priority;
    float* scores;             // or null: 0 for every vertex of a graph finished here (what the per-step path reports at the end)
    int32_t* rounds;           // += steps run here (or null)
    double* totals;            // += weight (priority without weights) that joined here (or null)
    int32_t* progress;         // += 1 per graph that decided something here (or null)
    int32_t* status;
    const unsigned long long* tail_word;  // this call's step kernels: tail_tag | largest number of undecided vertices of an active graph
    unsigned long long tail_tag;
    int32_t num_layers;
    int32_t diag;  // DGCN_DIAG builds only (tools/tail_probe.py): bit 0 no aggregation, bit 1 no transform, bit 2 no weight fetch, bit 3 no greedy-step ranks
    TailLayer layers[kTailMaxLayers];
};

// A row's aggregation for the two 16-byte chunks (features 4 kq .. and 16 + 4 kq ..) a lane owns: float32 fmaf chains entry by
// entry in storage order, or (layer index 0) fma chains in double rounded once after "+ Z0 (+ bias)" - fused.hip's RowAcc.
// Entries in groups of four: the columns of a group are one 4-byte LDS read, their values and the eight gathers are in flight
// together, the chain order is untouched.
// The first four entries of the row come in registers (us0 / av0: read once per step, not once per layer - most rows of a
// 64-vertex residual graph have no more), so a layer's aggregation is ONE LDS round trip: the gathers go out at once.
template <bool F64>
__device__ __forceinline__ void tail_row(const float* cvr, const uint8_t* cur, int cnt, unsigned us0, const float (&av0)[4], const float* zb,
                                         int kq, float4& yA, float4& yB, bool has_bias, float4 bA, float4 bB, int act) {
    float4 accA = make_float4(0.f, 0.f, 0.f, 0.f), accB = accA;
    double dA[4] = {0.0, 0.0, 0.0, 0.0}, dB[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = 0; i0 < cnt; i0 += 4) {
        const unsigned us = i0 == 0 ? us0 : *reinterpret_cast<const unsigned*>(cur + i0);  // (rows are 4-byte aligned and padded: kNbStride)
        float av[4];
        float4 qA[4], qB[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int u = (int)((us >> (8 * e)) & 0xffu) & (kTailMax - 1);  // (past the row's end: any valid row, not used)
            av[e] = i0 == 0 ? av0[e] : cvr[i0 + e];
            qA[e] = *reinterpret_cast<const float4*>(zb + u * kHid + ((kq ^ keyB(u)) << 2));
            qB[e] = *reinterpret_cast<const float4*>(zb + u * kHid + (((kq + 4) ^ keyB(u)) << 2));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (i0 + e < cnt) {
                if constexpr (F64) {
                    const double ad = (double)av[e];
                    dA[0] = fma(ad, (double)qA[e].x, dA[0]); dA[1] = fma(ad, (double)qA[e].y, dA[1]);
                    dA[2] = fma(ad, (double)qA[e].z, dA[2]); dA[3] = fma(ad, (double)qA[e].w, dA[3]);
                    dB[0] = fma(ad, (double)qB[e].x, dB[0]); dB[1] = fma(ad, (double)qB[e].y, dB[1]);
                    dB[2] = fma(ad, (double)qB[e].z, dB[2]); dB[3] = fma(ad, (double)qB[e].w, dB[3]);
                } else {
                    accA = fma4(av[e], qA[e], accA);
                    accB = fma4(av[e], qB[e], accB);
                }
            }
        }
    }
    float4 oA, oB;
    if constexpr (F64) {
        double a0 = (double)yA.x + dA[0], a1 = (double)yA.y + dA[1], a2 = (double)yA.z + dA[2], a3 = (double)yA.w + dA[3];
        double b0 = (double)yB.x + dB[0], b1 = (double)yB.y + dB[1], b2 = (double)yB.z + dB[2], b3 = (double)yB.w + dB[3];
        if (has_bias) {
            a0 += (double)bA.x; a1 += (double)bA.y; a2 += (double)bA.z; a3 += (double)bA.w;
            b0 += (double)bB.x; b1 += (double)bB.y; b2 += (double)bB.z; b3 += (double)bB.w;
        }
        oA = make_float4((float)a0, (float)a1, (float)a2, (float)a3);
        oB = make_float4((float)b0, (float)b1, (float)b2, (float)b3);
    } else {
        oA = make_float4(yA.x + accA.x, yA.y + accA.y, yA.z + accA.z, yA.w + accA.w);
        oB = make_float4(yB.x + accB.x, yB.y + accB.y, yB.z + accB.z, yB.w + accB.w);
        if (has_bias) {
            oA.x += bA.x; oA.y += bA.y; oA.z += bA.z; oA.w += bA.w;
            oB.x += bB.x; oB.y += bB.y; oB.z += bB.z; oB.w += bB.w;
        }
    }
    yA = make_float4(apply_act(oA.x, act), apply_act(oA.y, act), apply_act(oA.z, act), apply_act(oA.w, act));
    yB = make_float4(apply_act(oB.x, act), apply_act(oB.y, act), apply_act(oB.z, act), apply_act(oB.w, act));
}

#ifdef DGCN_DIAG
#define TAIL_DIAG(a, bit) (((a).diag >> (bit)) & 1)
#else
#define TAIL_DIAG(a, bit) 0
#endif

constexpr int kNbStride = 68;  // bytes per row of the neighbour / column lists (17 words: lanes = rows fall on distinct banks)
constexpr int kCvStride = 65;  // floats per row of the entry values

__global__ __launch_bounds__(kTailBlock) void k_tail(TailArgs a) {
    __shared__ __attribute__((aligned(16))) float bufB2[2 * kTailMax * kHid];  // Z1 of even / odd layers: a wave may write the next
                                                                               // layer's rows while others still gather this one's
    __shared__ __attribute__((aligned(16))) float bufA[kTailMax * kHid];  // H -> Z0 -> H'
    __shared__ __attribute__((aligned(16))) float cv[kTailMax * kCvStride];                            // entry values of the step (diagonal first)
    __shared__ __attribute__((aligned(16))) uint8_t cu[kTailMax * kNbStride];                          // ... and their columns (slots)
    __shared__ uint8_t nb[kTailMax * kNbStride];                          // neighbours among the graph's tail vertices, storage order
    __shared__ double wraw[kTailMax], pr[kTailMax], dinv[kTailMax], cand[kTailMax];
    __shared__ int orig[kTailMax], nbc[kTailMax], cnt[kTailMax];
    __shared__ unsigned long long adjs[kTailMax];
    __shared__ float zl[kTailMax];
    __shared__ uint8_t st[kTailMax];
    __shared__ int wsum[kTailBlock / 64];
    // read once per launch, used by every step: d^-1/2 of the degrees a 64-vertex graph can have, the last layer's two weight
    // columns, the first layer's weights while they are few (the reference's models: one input feature)
    __shared__ double dtab[kTailMax];
    __shared__ __attribute__((aligned(16))) float wlast[2 * kHid];
    constexpr int kW0Rows = 8;
    __shared__ __attribute__((aligned(16))) float w0s[kW0Rows * 2 * kHid];

    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Not before EVERY graph that is still being searched is small (see dgcn_solve_residual_batch): one load decides for the
    // whole launch.  (A word written by another call - none of this call's graphs was active - means there is nothing to do.)
    {
        const unsigned long long w = *a.tail_word;
        if ((w >> 32) > (a.tail_tag >> 32) && blockIdx.x == 0 && threadIdx.x == 0)  // no call has this number yet: whatever the
            *const_cast<unsigned long long*>(a.tail_word) = 0ull;                       // workspace held before; from the next call on the word works
        if ((w >> 32) != (a.tail_tag >> 32) || (unsigned)(w & 0xffffffffull) > (unsigned)kTailMax) return;
    }
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1], ng = n1 - n0;
    if (ng <= 0) return;

    // ---- who is undecided (ascending index: every "lower index first" rule holds on the slots)
    int carry = 0;
    for (int base = 0; base < ng; base += kTailBlock) {
        const int v = base + tid;
        const bool al = v < ng && a.state[n0 + v] == 0;
        const unsigned long long m = __ballot(al);
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int before = carry, tot = 0;
#pragma unroll
        for (int w = 0; w < kTailBlock / 64; ++w) {
            if (w < wave) before += wsum[w];
            tot += wsum[w];
        }
        const int pos = before + __popcll(m & ((1ull << lane) - 1ull));
        if (al && pos < kTailMax) orig[pos] = v;
        carry += tot;
        __syncthreads();
        if (carry > kTailMax) return;  // (uniform) not yet: the per-step kernels keep this graph
    }
    const int na0 = carry;
    if (na0 == 0) return;
    const bool hasw = a.weights != nullptr;
    if (tid < kTailMax) {
        st[tid] = tid < na0 ? 0 : 3;
        wraw[tid] = (tid < na0 && hasw) ? a.weights[n0 + orig[tid]] : 1.0;
        dtab[tid] = tid < a.table_len ? a.dinv_table[tid] : 0.0;
        wlast[tid] = a.layers[a.num_layers - 1].W[tid];  // [32][2]
    }
    const bool w0_cached = a.layers[0].cin <= kW0Rows;
    if (w0_cached)
        for (int i = tid; i < a.layers[0].cin * 2 * kHid; i += kTailBlock) w0s[i] = a.layers[0].W[i];
    __syncthreads();

    // ---- the adjacency among these vertices, once: four lanes per row, storage order kept by ballot ranks
    int fault = 0;
    {
        const int s = tid >> 2, sub = tid & 3, grp = lane >> 2;
        int rs = 0, re = 0;
        const int vo = s < na0 ? orig[s] : -1;
        if (s < na0) { rs = a.row_ptr[n0 + vo]; re = a.row_ptr[n0 + vo + 1]; }
        int base = 0;
        unsigned long long mask = 0ull;
        for (int j0 = rs; __any(j0 < re); j0 += 4) {
            const int j = j0 + sub;
            int t = -1;
            if (j < re) {
                const int u = a.col_idx[j] - n0;
                if (u < 0 || u >= ng) fault |= DGCN_FAULT_BAD_COLUMN;
                else if (u == vo) fault |= DGCN_FAULT_SELF_LOOP;
                else {
                    int lo = 0, hi = na0;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (orig[mid] < u) lo = mid + 1; else hi = mid;
                    }
                    if (lo < na0 && orig[lo] == u) t = lo;
                }
            }
            const bool keep = t >= 0;
            const unsigned bits = (unsigned)(__ballot(keep) >> (grp * 4)) & 0xfu;
            if (keep) {
                const int slot = base + __popc(bits & ((1u << sub) - 1u));
                if (slot < kTailMax) nb[s * kNbStride + slot] = (uint8_t)t;  // (more entries than vertices: duplicate columns - flagged below)
                mask |= 1ull << t;
            }
            base += __popc(bits);
        }
        mask |= __shfl_xor(mask, 1);
        mask |= __shfl_xor(mask, 2);
        if (base >= kTailMax) { fault |= DGCN_FAULT_DEGREE_RANGE; base = kTailMax - 1; }
        if (sub == 0 && s < kTailMax) { adjs[s] = s < na0 ? mask : 0ull; nbc[s] = s < na0 ? base : 0; }
    }
    __syncthreads();
    const unsigned long long myadj = adjs[lane];  // lane = slot, in every wave
    const double wv = wraw[lane];
    const double wq = hasw ? wv : 0.0;            // what counts in totals / rollout completions (fused.hip: wl)

    const int L = a.num_layers;
    int steps = 0;
    double added = 0.0;  // (thread 0)
    bool nan_stop = false;
    float bfrag[8][4];

    for (;;) {
        const bool al = st[lane] == 0;
        const unsigned long long alive = __ballot(al);
        // nothing left, or no positive weight left (np.sum(wts_nn) <= 0 -> break, mwis_gdpg_call.py:286)
        if (__ballot(al && wv > 0.0) == 0ull) break;
        double wmax = al ? wv : -1.0 / 0.0;
        if (a.feature_mode == 1)
            for (int off = 1; off < 64; off <<= 1) wmax = fmax(wmax, __shfl_xor(wmax, off));

        // ---- the residual graph's support: degrees, d^-1/2, entries (diagonal first, then the row in storage order)
        if (wave == 0 && al) {
            int d = 0;
            const int c = nbc[lane];
            for (int i = 0; i < c; ++i) d += st[nb[lane * kNbStride + i]] == 0;
            double dv = 0.0;
            if (d < a.table_len) dv = dtab[d]; else fault |= DGCN_FAULT_DEGREE_RANGE;  // (d <= 63)
            dinv[lane] = dv;
        }
        __syncthreads();
        if (wave == 0 && al) {
            const double dv = dinv[lane];
            const int c = nbc[lane];
            cu[lane * kNbStride] = (uint8_t)lane;
            cv[lane * kCvStride] = 1.0f;  // (I - A_hat)[v][v], zero-diagonal adjacency
            int k = 1;
            for (int i = 0; i < c; ++i) {
                const int u = nb[lane * kNbStride + i];
                if (st[u] == 0) {
                    cu[lane * kNbStride + k] = (uint8_t)u;
                    cv[lane * kCvStride + k] = (float)(-(dinv[u] * dv));  // reference order, float64, then the float32 feed cast
                    ++k;
                }
            }
            cnt[lane] = k;
        }
        // ---- layer 0: transform on the VALU (any input width), one thread per (row, 16 outputs)
        {
            const TailLayer& L0 = a.layers[0];
            const int v = tid >> 2, c0 = 16 * (tid & 3);
            if (st[v] == 0) {
                const float x = a.feature_mode == 1 ? (float)(wraw[v] / (wmax + 1e-9)) : a.x_const;  // mwis_gdpg_call.py:88
                float acc[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                if (w0_cached) {
                    for (int k = 0; k < L0.cin; ++k) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[i] = fmaf(x, w0s[k * 2 * kHid + c0 + i], acc[i]);
                    }
                } else {
                    for (int k = 0; k < L0.cin; ++k) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[i] = fmaf(x, L0.W[k * 2 * kHid + c0 + i], acc[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; i += 4) {
                    float* dst = c0 < kHid ? bufA + swz(v, c0 + i) : bufB2 + swzB(v, c0 - kHid + i);
                    *reinterpret_cast<float4*>(dst) = make_float4(acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
                }
            }
        }
        // Layers 0 .. L - 2.  A wave's transform tile and its aggregation rows are the SAME sixteen rows (tile = wave, four lanes per
        // row), so H / Z0 / H' in bufA never cross a wave: only Z1 does, and with Z1 double-buffered ONE workgroup barrier per
        // layer remains (between the transforms' Z1 writes and the gathers).  Weights and bias of a layer are requested a
        // whole layer ahead (two fragment sets in registers).
        const int v = tid >> 2, kq = tid & 3;
        const bool mine = st[v] == 0;
        float bnext[8][4];
        float4 bA = make_float4(0.f, 0.f, 0.f, 0.f), bB = bA, nA = bA, nB = bA;
        if (a.layers[0].bias) {
            bA = *reinterpret_cast<const float4*>(a.layers[0].bias + 4 * kq);
            bB = *reinterpret_cast<const float4*>(a.layers[0].bias + 4 * (kq + 4));
        }
        __syncthreads();
        // this row's entry count and first four entries, for every layer of the step
        int rc = 0;
        unsigned us0 = 0u;
        float av0[4] = {0.f, 0.f, 0.f, 0.f};
        if (mine) {
            rc = cnt[v];
            us0 = *reinterpret_cast<const unsigned*>(cu + v * kNbStride);
#pragma unroll
            for (int e = 0; e < 4; ++e) av0[e] = cv[v * kCvStride + e];
        }
        for (int l = 0; l + 1 < L; ++l) {
            const TailLayer& Ll = a.layers[l];
            float* zb = bufB2 + (l & 1) * (kTailMax * kHid);
            if (l + 2 < L) {  // the next hidden layer's fragments and bias: a layer's worth of time to arrive
                load_bfrag(a.layers[l + 1].W, bnext, TAIL_DIAG(a, 2), l == 0);  // (layer index 1: the f64 MFMA's lane map)
                if (a.layers[l + 1].bias) {
                    nA = *reinterpret_cast<const float4*>(a.layers[l + 1].bias + 4 * kq);
                    nB = *reinterpret_cast<const float4*>(a.layers[l + 1].bias + 4 * (kq + 4));
                }
            }
            if (l >= 1) {
                if (TAIL_DIAG(a, 1)) {}
                else if (l == 1) hidden_transform_f64<kTailBlock>(bfrag, kTailMax, bufA, zb);
                else hidden_transform<kTailBlock>(bfrag, kTailMax, bufA, zb);
                __syncthreads();  // (on gfx950 a workgroup barrier waits for the wave's LDS operations only: the fragments requested above stay in flight)
            }
            if (mine && !TAIL_DIAG(a, 0)) {
                const int c = rc;
                float4* ownA = reinterpret_cast<float4*>(bufA + v * kHid + ((kq ^ (v & 7)) << 2));
                float4* ownB = reinterpret_cast<float4*>(bufA + v * kHid + (((kq + 4) ^ (v & 7)) << 2));
                float4 yA = *ownA, yB = *ownB;
                if (l == 0) tail_row<true>(cv + v * kCvStride, cu + v * kNbStride, c, us0, av0, zb, kq, yA, yB, Ll.bias != nullptr, bA, bB, Ll.act);
                else tail_row<false>(cv + v * kCvStride, cu + v * kNbStride, c, us0, av0, zb, kq, yA, yB, Ll.bias != nullptr, bA, bB, Ll.act);
                *ownA = yA;
                *ownB = yB;
            }
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) bfrag[s8][ct] = bnext[s8][ct];
            bA = nA;
            bB = nB;
        }
        __syncthreads();
        // ---- last layer (32 -> 1): two k-ordered chains per row, then the width-1 aggregation
        float score = 0.f, z0 = 0.f;
        {
            const TailLayer& LL = a.layers[L - 1];
            if (wave == 0 && al) {
                float z1 = 0.f;
#pragma unroll
                for (int c = 0; c < kHid / 4; ++c) {
                    const float4 h = *reinterpret_cast<const float4*>(bufA + lane * kHid + ((c ^ (lane & 7)) << 2));
                    const float hk[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        z0 = fmaf(hk[i], wlast[(4 * c + i) * 2 + 0], z0);
                        z1 = fmaf(hk[i], wlast[(4 * c + i) * 2 + 1], z1);
                    }
                }
                zl[lane] = z1;
            }
            __syncthreads();
            if (wave == 0 && al) {
                float acc = 0.f;
                const int c = cnt[lane];
                for (int i = 0; i < c; ++i) acc = fmaf(cv[lane * kCvStride + i], zl[cu[lane * kNbStride + i]], acc);
                float o = z0 + acc;
                if (LL.bias) o += LL.bias[0];
                score = apply_act(o, LL.act);
                double p = (double)score;  // mwis_gdpg_call.py:211-216: float32 x float64 -> float64
                if (a.predict_mwis && hasw) p *= wv;
                pr[lane] = p;
            }
            __syncthreads();
        }
        // ---- the greedy step, on ballots; every wave holds the same per-slot values (lane = slot)
        const double p = pr[lane];
        if (__ballot(al && p != p) != 0ull) {  // the reference's argmax / argsort would act on the NaN: report, leave the graph alone
            nan_stop = true;
            break;
        }
        int gk = 0, wk = 0;  // ranks among the undecided under (priority desc, index asc) and (weight desc, index asc)
        for (unsigned long long m = alive; m; m &= m - 1ull) {
            const int u = __ffsll((long long)m) - 1;  // (uniform)
            const double pu = pr[u], wu = hasw ? wraw[u] : 0.0;
            gk += (pu > p) || (pu == p && u < lane);
            wk += (wu > wq) || (wu == wq && u < lane);
        }
        const int rk = (a.greedy_mode == 2 && !a.by_priority) ? wk : gk;  // the order the local greedy search runs under
        unsigned long long ahead = 0ull;  // live-or-not neighbours that come before this vertex in that order
        for (unsigned long long m = alive; m; m &= m - 1ull) {
            const int u = __ffsll((long long)m) - 1;
            const int ru = __shfl(rk, u);
            if (((myadj >> u) & 1ull) && ru < rk) ahead |= 1ull << u;
        }
        if (a.greedy_mode == 0) {
            // `max_rounds` synchronous rounds of the local greedy search (solve_mwis_dit: one)
            unsigned long long live = alive, joined = 0ull, killed = 0ull;
            int rounds = 0;
            while (live) {
                const bool lv = (live >> lane) & 1ull;
                const bool won = lv && (ahead & live) == 0ull;
                const unsigned long long W = __ballot(won);
                const unsigned long long K = __ballot(lv && !won && (myadj & W) != 0ull);
                joined |= W;
                killed |= K;
                live &= ~(W | K);
                ++rounds;
                if (a.max_rounds > 0 && rounds >= a.max_rounds) break;
            }
            double part = ((joined >> lane) & 1ull) ? (hasw ? wv : p) : 0.0;
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (wave == 0) {
                if ((joined >> lane) & 1ull) st[lane] = 1;
                else if ((killed >> lane) & 1ull) st[lane] = 2;
            }
            if (tid == 0) added += part;
        } else if (a.greedy_mode == 1) {
            // the global best joins (np.argmax: lowest index among equals), its neighbours leave
            const int c = __ffsll((long long)__ballot(al && gk == 0)) - 1;
            const unsigned long long adjc = __shfl(myadj, c);
            if (wave == 0) {
                if (lane == c) st[lane] = 1;
                else if (al && ((adjc >> lane) & 1ull)) st[lane] = 2;
            }
            const double pc = __shfl(hasw ? wv : p, c);
            if (tid == 0) added += pc;
        } else {
            // rollout: the first `beam` undecided vertices in priority order; each is completed by the local greedy search
            // (by weight, or by priority) on the residual graph minus its closed neighbourhood; the best total joins
            int nc = __popcll(alive);
            nc = min(nc, min(a.beam, kTailMax));
            // lane i takes the value of the i-th undecided slot (the fused kernel sums the completion's weights over the
            // renumbered vertices with the same butterfly)
            int src = -1;
            {
                int k = 0;
                for (unsigned long long m = alive; m; m &= m - 1ull, ++k)
                    if (lane == k) src = __ffsll((long long)m) - 1;
            }
            for (int i = wave; i < nc; i += kTailBlock / 64) {
                const int c = __ffsll((long long)__ballot(al && gk == i)) - 1;
                const unsigned long long adjc = __shfl(myadj, c);
                unsigned long long live = alive & ~(adjc | (1ull << c));
                double mine = 0.0;
                while (live) {
                    const bool lv = (live >> lane) & 1ull;
                    const bool won = lv && (ahead & live) == 0ull;
                    const unsigned long long W = __ballot(won);
                    const unsigned long long K = __ballot(lv && !won && (myadj & W) != 0ull);
                    if (won) mine = wq;
                    live &= ~(W | K);
                }
                const double moved = __shfl(mine, src < 0 ? lane : src);
                double tot = src < 0 ? 0.0 : moved;
                for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
                const double wc = __shfl(wq, c);
                if (lane == 0) cand[i] = wc + tot;
            }
            __syncthreads();
            // np.isclose(cand, cand.max(), rtol=1e-12, atol=0): the first candidate within tolerance wins (none within it - totals
            // that are not finite -: the first).  A lane per candidate, every wave for itself (no walk of thread 0 over the list,
            // no barrier behind it).
            const double cvl = lane < nc ? cand[lane] : -1.0 / 0.0;
            const double cmx = wave_max_f64(cvl);
            const unsigned long long tied = __ballot(lane < nc && (cvl == cmx || fabs(cvl - cmx) <= 1e-12 * fabs(cmx)));  // (equal infinities are close)
            const int best = tied ? __ffsll((long long)tied) - 1 : 0;
            const int c = __ffsll((long long)__ballot(al && gk == best)) - 1;
            const unsigned long long adjc = __shfl(myadj, c);
            if (wave == 0) {
                if (lane == c) st[lane] = 1;
                else if (al && ((adjc >> lane) & 1ull)) st[lane] = 2;
            }
            const double wc = __shfl(wq, c);
            if (tid == 0) added += wc;
        }
        ++steps;
        __syncthreads();
    }

    // ---- out: states of the vertices this launch decided, what the caller's arrays report for the graph
    if (tid < na0 && st[tid] != 0) a.state[n0 + orig[tid]] = st[tid];
    if (a.scores && !nan_stop)  // (a faulting graph: its results are invalid, the status word says so; scores stay as they were)
        for (int v = tid; v < ng; v += kTailBlock) a.scores[n0 + v] = 0.f;
    if (tid == 0) {
        if (nan_stop) {
            atomicOr(a.status, DGCN_FAULT_NAN_PRIORITY);
            if (a.rounds) a.rounds[g] = -1;
            if (a.totals) a.totals[g] = 0.0;
        } else {
            if (a.rounds) a.rounds[g] += steps;
            if (a.totals) a.totals[g] += added;
        }
        if (steps > 0 && a.progress) atomicAdd(a.progress, 1);
    }
    if (fault) atomicOr(a.status, fault);
}

// Which (model, call) pairs the tail takes: [I, L] stacks F -> 32 -> ... -> 32 -> 1 of at least three layers (the shapes k_big
// takes), constant or weight-derived input features, scores computed here.
int tail_takes(const DgcnModel* m, const float* X, int32_t options) {
    const bool off = opt(OPT_TAIL) == 0;
    if (off || !m || !m->layers_host || m->num_supports != 2 || X || (options & DGCN_RESIDUAL_SCORES_GIVEN)) return 0;
    const int L = m->num_layers;
    if (L < 3 || L > kTailMaxLayers) return 0;
    for (int l = 0; l < L; ++l) {
        const DgcnLayer& Ly = m->layers_host[l];
        if (!Ly.weights) return 0;
        if (l == 0 && (Ly.in_dim < 1 || Ly.in_dim > 64)) return 0;
        if (l > 0 && Ly.in_dim != kHid) return 0;
        if (Ly.out_dim != (l == L - 1 ? 1 : kHid)) return 0;
        if (Ly.bias && (reinterpret_cast<uintptr_t>(Ly.bias) & 15)) return 0;  // (k_tail reads a hidden layer's bias as float4, like k_big)
    }
    return 1;
}

int tail_finish(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, float x_const,
                int32_t feature_mode, const double* weights, int32_t predict_mwis, int32_t greedy_mode, int32_t max_rounds,
                int32_t beam, int32_t options, float* scores, uint8_t* state, int32_t* rounds, double* totals, int32_t* progress,
                int32_t* status, hipStream_t s, const unsigned long long* tail_word, unsigned long long tail_tag) {
    TailArgs a = {};
    a.tail_word = tail_word; a.tail_tag = tail_tag;
    a.graph_ptr = b->graph_ptr; a.row_ptr = b->row_ptr; a.col_idx = b->col_idx;
    a.state = state; a.weights = weights; a.dinv_table = dinv_table; a.table_len = table_len;
    a.feature_mode = feature_mode; a.x_const = x_const; a.predict_mwis = predict_mwis; a.greedy_mode = greedy_mode;
    a.max_rounds = max_rounds; a.beam = beam; a.by_priority = (options & DGCN_RESIDUAL_COMPLETE_BY_PRIORITY) ? 1 : 0;
    a.scores = scores; a.rounds = rounds; a.totals = totals; a.progress = progress; a.status = status;
    a.num_layers = m->num_layers;
    for (int l = 0; l < m->num_layers; ++l) {
        const DgcnLayer& Ly = m->layers_host[l];
        a.layers[l].W = Ly.weights; a.layers[l].bias = Ly.bias; a.layers[l].cin = Ly.in_dim; a.layers[l].cout = Ly.out_dim;
        a.layers[l].act = Ly.act; a.layers[l].pad = 0;
    }
#ifdef DGCN_DIAG
    a.diag = opt(OPT_DIAG_FLAGS);
#endif
    TimedLaunch t("tail_finish", s);
    DGCN_LAUNCH(t, k_tail, dim3((unsigned)b->num_graphs), dim3(kTailBlock), 0, s, a);
    return check_launch("k_tail");
}

}  // namespace dgcn
