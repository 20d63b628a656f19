// One-layer models (F -> 1) - and, further down, two-layer models (F -> C -> 1) on constant features - on graphs of ANY size the any-size path takes (<= 9 600 vertices), adjacency in, set out, ONE launch:
// what the reference's multi-channel launchers actually run - `--num_layer=1 --num_channels=3` on the joint conflict graph of
// K * nflows vertices (bash/twc_major_wireless_mc_test.sh:3,6,9, bash/test_wireless_gcn_rollout.sh:6-8,
// wireless_dqn_test_mc.py:161, 244-289) - and every residual step of solve_mwis_dit / _cit / the rollouts on such a model
// (mwis_gdpg_call.py:278-318, 343-384, 596-659).  shallow.hip serves the same models up to 512 vertices with the graph's
// entry list in LDS; beyond that these shapes used to run as k_supports -> layer-by-layer forward -> k_lgs (three to five
// launches), a residual step as seven to nine.
//
// For one layer the score is a closed form in the (residual) degrees: with L = I - D^-1/2 A D^-1/2,
//     score_v = act( z0_v + [ 1.z1_v + sum_u (float)(-(d_u^-1/2 d_v^-1/2)) z1_u ] + b ),   z = x.[w0 | w1]
// - no 32-wide image, no records, no MFMA: d^-1/2 per vertex in LDS, one chain per vertex.  Arithmetic exactly as shallow.hip /
// the twin (oracle/dgcn_oracle.c: orc_spmm_f64 - layer index 0 carries its row chain in double): the transform is a float32
// fmaf chain over the input features, the row sum ONE fma chain in double over [diagonal, the row's (undecided) neighbours in
// CSR order], score = act((float)((double)z0 + acc [+ (double)b])), priority (double)score * weight (mwis_dqn_call.py:232).
//
// One 1 024-thread workgroup per graph, vertex v = thread + 1 024 p.  LDS is k_lgs's carve-up (lgs.hip) - float64 priorities,
// row offsets, state bytes, the graph's columns as 16-bit local ids when they fit - with two of its arrays doing double duty
// before the search starts: the priorities' space holds d^-1/2 (float64) while the chains run, the row offsets' space holds
// z1 per vertex (explicit / weight-derived features) or else the float32 scores on their way to becoming priorities.
//   residual = 0  dgcn_solve_batch: every vertex takes part (heuristics.py:77-116 on gcn_wts)
//   residual = 1  one step of dgcn_solve_residual_batch: `state` is the running state (0 = undecided); degrees, chains and the
//                 greedy step see the undecided vertices only - the re-sliced graph of mwis_gdpg_call.py:284-285 without
//                 re-slicing it: a chain that skips decided neighbours IS the chain over the re-sliced row (same order);
//                 a graph with nothing left or no positive weight left is left alone (np.sum(wts_nn) <= 0 -> break, :286)
//   mode 0 local greedy rounds (max_rounds; lgs_rounds.h), 1 the best-priority vertex joins (solve_mwis_cit), 2 scores and
//   priorities only (the rollout's candidate / completion launches of general.hip follow)
// Bound: latency (dependent LDS round trips of the rounds); HBM sees the adjacency once or twice (columns that do not fit the
// LDS are re-read from L2 by every sweep), `nnz * 4 + N * 20` bytes in, `N + 16` out per graph - shallow.hip's figure.
#include <algorithm>
#include <atomic>

#include "common.h"
#include "lgs_rounds.h"
#include "cand_select.h"
#include "rollout_bits.h"

namespace dgcn {

constexpr int kWideBlock = 1024;
constexpr int kWideMaxNodes = 9600;  // the search's LDS state (lgs.hip)

struct WideArgs {
    const int32_t* graph_ptr;
    const int32_t* row_ptr;
    const int32_t* col_idx;
    const double* dinv;        // float64 d^-1/2 by degree
    int32_t table_len;
    const float* X;            // [num_nodes][cin] or null
    float x_const;
    int32_t cin, feature_mode; // feature_mode 1: x = weight / (largest undecided weight + 1e-9) (mwis_gdpg_call.py:88)
    const float* W;            // [cin][2] (two layers: [cin][2 C])
    const float* bias;         // [1] or null (two layers: [C])
    int32_t act;
    // two-layer models (F -> C -> 1, C <= 64, constant input features): the second layer
    int32_t two, C;
    const float* W2;           // [C][2]
    const float* bias2;        // [1] or null
    int32_t act2;
    const double* weights;     // or null
    int32_t predict_mwis, residual, scores_given, mode, max_rounds;
    float* sc;                 // [num_nodes] scores, original numbering: out (in with scores_given); never null
    uint8_t* state;
    int32_t* rounds;
    double* totals;
    int32_t* status;
    int32_t* progress;
    double* prio;              // mode 2: [num_nodes] out
    int32_t* active;           // mode 2: [num_graphs] out
    int32_t* cid;              // mode 2: [num_graphs][64] the rollout's candidates (cand_select.h), or null (k_res_cand follows)
    int32_t beam;
    int32_t ahead_rounds;      // mode 0, whole searches: the rounds on ahead lists (lgs_rounds_ahead, lgs_rounds.h); 0: lgs_rounds.h's (option "wide_ahead" = 0)
    int32_t roll_off;          // mode 2: byte offset of the LDS the completions and the pick run in (rollout_bits.h) - the whole step in
    int32_t by_priority;       // this launch; 0: general.hip's k_lgs / k_res_pick launches follow.  by_priority: the completions' order
    unsigned long long* tail_word;
    unsigned long long tail_tag;
    int32_t max_nodes, cols_cap;
};

__host__ __device__ __forceinline__ size_t wide_pad16(int x) { return (size_t)((x + 15) & ~15); }

static size_t wide_lds_bytes(int max_nodes, int cols_cap) {
    const size_t ro = (size_t)((max_nodes + 1 + 3) & ~3) * 4;
    return (size_t)max_nodes * 8 + 1024 * 8 + 4 * 8 + ro + 2 * wide_pad16(max_nodes + 1) + (size_t)cols_cap * 2;
}

// the column of entry j of the graph (local id; anything >= ng is not a vertex of the graph)
template <bool COLS_LDS>
__device__ __forceinline__ int wide_col(const uint16_t* cl, const int32_t* cg, int j, int e0, int n0) {
    if (COLS_LDS) return (int)cl[j - e0];
    return cg[j] - n0;
}

template <bool COLS_LDS>
__device__ __forceinline__ void wide_scores(const WideArgs& a, int n0, int ng, int e0, const uint8_t* st, const uint16_t* cl, double* dinv,
                                            float* z1, float* stash, double wmax, int& fault) {
    const bool need_z1 = z1 != nullptr;
    const float bias = a.bias ? a.bias[0] : 0.f;
    // ---- (residual) degrees -> d^-1/2 per vertex; z1 per vertex where the features differ from vertex to vertex
    for (int v = threadIdx.x; v < ng; v += kWideBlock) {
        if (st[v] != 0) continue;
        const int rs = a.row_ptr[n0 + v], re = a.row_ptr[n0 + v + 1];
        int deg = re - rs;
        if (a.residual) {
            deg = 0;
#pragma unroll 4
            for (int j = rs; j < re; ++j) {
                const int u = wide_col<COLS_LDS>(cl, a.col_idx, j, e0, n0);
                if ((unsigned)u >= (unsigned)ng) { fault |= DGCN_FAULT_BAD_COLUMN; continue; }
                deg += st[u] == 0;
            }
        }
        double dv = 0.0;
        if (deg < a.table_len) dv = a.dinv[deg]; else fault |= DGCN_FAULT_DEGREE_RANGE;
        dinv[v] = dv;
        if (need_z1) {
            float q1 = 0.f;
            if (a.feature_mode == 1) {
                const float f = (float)((a.weights ? a.weights[n0 + v] : 1.0) / (wmax + 1e-9));
                for (int k = 0; k < a.cin; ++k) q1 = fmaf(f, a.W[k * 2 + 1], q1);
            } else {
                for (int k = 0; k < a.cin; ++k) q1 = fmaf(a.X[(size_t)(n0 + v) * a.cin + k], a.W[k * 2 + 1], q1);
            }
            z1[v] = q1;
        }
    }
    __syncthreads();
    // ---- one chain per undecided vertex
    for (int v = threadIdx.x; v < ng; v += kWideBlock) {
        float s = 0.f;  // what a decided vertex reports (general.hip: k_res_scatter)
        if (st[v] == 0) {
            float z0 = 0.f, z1v = 0.f;
            if (a.feature_mode == 1) {
                const float f = (float)((a.weights ? a.weights[n0 + v] : 1.0) / (wmax + 1e-9));
                for (int k = 0; k < a.cin; ++k) { z0 = fmaf(f, a.W[k * 2], z0); z1v = fmaf(f, a.W[k * 2 + 1], z1v); }
            } else {
                for (int k = 0; k < a.cin; ++k) {
                    const float x = a.X ? a.X[(size_t)(n0 + v) * a.cin + k] : a.x_const;
                    z0 = fmaf(x, a.W[k * 2], z0);
                    z1v = fmaf(x, a.W[k * 2 + 1], z1v);
                }
            }
            const double dvv = dinv[v];
            const double zd = (double)z1v;
            double acc = fma(1.0, zd, 0.0);  // the diagonal entry of L comes first
            const int rs = a.row_ptr[n0 + v], re = a.row_ptr[n0 + v + 1];
            int j = rs;
            for (; j + 4 <= re; j += 4) {  // four entries' loads in flight, chain order unchanged
                int u[4];
                double du[4], zu[4];
                bool keep[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = wide_col<COLS_LDS>(cl, a.col_idx, j + i, e0, n0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    keep[i] = (unsigned)u[i] < (unsigned)ng;
                    if (!keep[i]) { fault |= DGCN_FAULT_BAD_COLUMN; u[i] = v; }
                    if (keep[i] && u[i] == v) fault |= DGCN_FAULT_SELF_LOOP;
                    if (a.residual) keep[i] = keep[i] && st[u[i]] == 0;
                    du[i] = dinv[u[i]];
                    zu[i] = need_z1 ? (double)z1[u[i]] : zd;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (keep[i]) acc = fma((double)(float)(-(du[i] * dvv)), zu[i], acc);
            }
            for (; j < re; ++j) {
                const int u = wide_col<COLS_LDS>(cl, a.col_idx, j, e0, n0);
                if ((unsigned)u >= (unsigned)ng) { fault |= DGCN_FAULT_BAD_COLUMN; continue; }
                if (u == v) fault |= DGCN_FAULT_SELF_LOOP;
                if (a.residual && st[u] != 0) continue;
                acc = fma((double)(float)(-(dinv[u] * dvv)), need_z1 ? (double)z1[u] : zd, acc);
            }
            acc = (double)z0 + acc;
            if (a.bias) acc += (double)bias;
            s = apply_act((float)acc, a.act);
        }
        a.sc[n0 + v] = s;
        if (stash) stash[v] = s;
    }
}

// ---- two-layer models F -> C -> 1 on constant input features (the reference's l = 2 launchers and checkpoints; C <= 64).  Layer 0's
// output is again a function of the vertex's (residual) neighbourhood's d^-1/2 only - every row of Z = x.[W0 | W1] is the same 2 C
// numbers (z0c | z1c, formed once per workgroup into LDS) - so H never exists as a matrix: a thread forms its vertex's H sixteen
// features at a time (the contract's double chains over [diagonal, undecided neighbours in CSR order]: orc_spmm_f64) and feeds
// them straight into the second layer's transform, two fma chains in double over k = 0 .. C - 1 (layer index 1: orc_transform_f64)
// - z0 goes to the scores array (this thread reads it back below), z1 into LDS.  Then the second layer's aggregation at width 1,
// a float32 fmaf chain over [diagonal, neighbours] (orc_spmm), + z0, + bias, activation.  Two walks over the adjacency.
template <bool COLS_LDS>
__device__ __forceinline__ void wide_scores2(const WideArgs& a, int n0, int ng, int e0, const uint8_t* st, const uint16_t* cl, double* dinv,
                                             float* z1, float* cst, int& fault) {
    const int C = a.C;
    float* z0c = cst;            // [C]
    float* z1c = cst + 64;       // [C]
    float* w2 = cst + 128;       // [C][2]
    float* b0 = cst + 256;       // [C]
    for (int k = threadIdx.x; k < C; k += kWideBlock) {
        float q0 = 0.f, q1 = 0.f;
        for (int kk = 0; kk < a.cin; ++kk) {
            q0 = fmaf(a.x_const, a.W[kk * 2 * C + k], q0);
            q1 = fmaf(a.x_const, a.W[kk * 2 * C + C + k], q1);
        }
        z0c[k] = q0;
        z1c[k] = q1;
        w2[2 * k] = a.W2[2 * k];
        w2[2 * k + 1] = a.W2[2 * k + 1];
        b0[k] = a.bias ? a.bias[k] : 0.f;
    }
    // (residual) degrees -> d^-1/2 per vertex
    for (int v = threadIdx.x; v < ng; v += kWideBlock) {
        if (st[v] != 0) continue;
        const int rs = a.row_ptr[n0 + v], re = a.row_ptr[n0 + v + 1];
        int deg = re - rs;
        if (a.residual) {
            deg = 0;
#pragma unroll 4
            for (int j = rs; j < re; ++j) {
                const int u = wide_col<COLS_LDS>(cl, a.col_idx, j, e0, n0);
                if ((unsigned)u >= (unsigned)ng) { fault |= DGCN_FAULT_BAD_COLUMN; continue; }
                deg += st[u] == 0;
            }
        }
        double dv = 0.0;
        if (deg < a.table_len) dv = a.dinv[deg]; else fault |= DGCN_FAULT_DEGREE_RANGE;
        dinv[v] = dv;
    }
    __syncthreads();
    // ---- first walk: layer 0 and the second layer's transform
    for (int v = threadIdx.x; v < ng; v += kWideBlock) {
        if (st[v] != 0) continue;
        const double dvv = dinv[v];
        const int rs = a.row_ptr[n0 + v], re = a.row_ptr[n0 + v + 1];
        double z0d = 0.0, z1d = 0.0;
        for (int c0 = 0; c0 < C; c0 += 16) {
            double zc[16], acc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                zc[k] = c0 + k < C ? (double)z1c[c0 + k] : 0.0;
                acc[k] = fma(1.0, zc[k], 0.0);  // the diagonal entry of L comes first
            }
            for (int j = rs; j < re; ++j) {
                const int u = wide_col<COLS_LDS>(cl, a.col_idx, j, e0, n0);
                if ((unsigned)u >= (unsigned)ng) { fault |= DGCN_FAULT_BAD_COLUMN; continue; }
                if (u == v) fault |= DGCN_FAULT_SELF_LOOP;
                if (a.residual && st[u] != 0) continue;
                const double vd = (double)(float)(-(dinv[u] * dvv));
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = fma(vd, zc[k], acc[k]);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (c0 + k < C) {
                    double d = (double)z0c[c0 + k] + acc[k];
                    if (a.bias) d += (double)b0[c0 + k];
                    const double h = (double)apply_act((float)d, a.act);
                    z0d = fma(h, (double)w2[2 * (c0 + k)], z0d);
                    z1d = fma(h, (double)w2[2 * (c0 + k) + 1], z1d);
                }
            }
        }
        z1[v] = (float)z1d;
        a.sc[n0 + v] = (float)z0d;  // (read back by this thread below)
    }
    __syncthreads();
    // ---- second walk: the second layer's aggregation at width 1
    const float bias2 = a.bias2 ? a.bias2[0] : 0.f;
    for (int v = threadIdx.x; v < ng; v += kWideBlock) {
        float s = 0.f;
        if (st[v] == 0) {
            const double dvv = dinv[v];
            const int rs = a.row_ptr[n0 + v], re = a.row_ptr[n0 + v + 1];
            float acc = fmaf(1.0f, z1[v], 0.f);
            int j = rs;
            for (; j + 4 <= re; j += 4) {  // four entries' loads in flight, chain order unchanged
                int u[4];
                bool keep[4];
                float val[4], zu[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = wide_col<COLS_LDS>(cl, a.col_idx, j + i, e0, n0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    keep[i] = (unsigned)u[i] < (unsigned)ng;
                    if (!keep[i]) u[i] = v;
                    if (a.residual) keep[i] = keep[i] && st[u[i]] == 0;
                    val[i] = (float)(-(dinv[u[i]] * dvv));
                    zu[i] = z1[u[i]];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (keep[i]) acc = fmaf(val[i], zu[i], acc);
            }
            for (; j < re; ++j) {
                const int u = wide_col<COLS_LDS>(cl, a.col_idx, j, e0, n0);
                if ((unsigned)u >= (unsigned)ng) continue;
                if (a.residual && st[u] != 0) continue;
                acc = fmaf((float)(-(dinv[u] * dvv)), z1[u], acc);
            }
            s = a.sc[n0 + v] + acc;
            if (a.bias2) s = s + bias2;
            s = apply_act(s, a.act2);
        }
        a.sc[n0 + v] = s;
    }
}

template <int LPV>
__global__ __launch_bounds__(kWideBlock) void k_wide1(WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wide_raw[];
    const int g = blockIdx.x;
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1], ng = n1 - n0;
    // carve (lgs.hip's, one more state slot): [f64 prio | f64 reduce[1024] | u64 acc[4] | i32 row offsets | u8 st | u8 nw | u16 cols]
    double* pr = reinterpret_cast<double*>(wide_raw);
    double* red = pr + a.max_nodes;
    unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(red + 1024);
    int* rol = reinterpret_cast<int*>(acc64 + 4);
    uint8_t* st = reinterpret_cast<uint8_t*>(rol + ((a.max_nodes + 1 + 3) & ~3));
    uint8_t* nw = st + wide_pad16(a.max_nodes + 1);
    uint16_t* cl = reinterpret_cast<uint16_t*>(nw + wide_pad16(a.max_nodes + 1));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (a.cid && threadIdx.x < kCandMaxBeam) a.cid[(size_t)g * kCandMaxBeam + threadIdx.x] = -1;  // (a graph left alone has no candidates)
    if (ng <= 0) {
        if (threadIdx.x == 0) {
            if (a.rounds) a.rounds[g] = 0;
            if (a.totals) a.totals[g] = 0.0;
            if (a.active) a.active[g] = 0;
        }
        return;
    }
    int fault = 0;
    const int e0 = a.row_ptr[n0], e1 = a.row_ptr[n1];
    const bool cols_lds = (e1 - e0) <= a.cols_cap;
    // ---- the running state; is anything left to do (residual steps); the graph's columns as 16-bit local ids
    int cnt = 0, pos = 0;
    double mx = -1.0 / 0.0;
    for (int v = threadIdx.x; v < ng; v += kWideBlock) {
        const uint8_t s0 = a.residual ? a.state[n0 + v] : (uint8_t)0;
        st[v] = s0;
        nw[v] = 0;
        if (a.residual) {
            const bool alive = s0 == 0;
            const double w = a.weights ? a.weights[n0 + v] : 1.0;
            cnt += alive;
            pos |= alive && w > 0.0;
            if (alive) mx = fmax(mx, w);
        }
    }
    if (threadIdx.x == 0) st[ng] = 3;  // the slot a column outside the graph points to: takes part in nothing
    if (cols_lds) {
        for (int base = e0 + threadIdx.x; base < e1; base += kWideBlock * 4) {  // four loads in flight per thread
            int c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = (base + i * kWideBlock < e1) ? a.col_idx[base + i * kWideBlock] : n0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (base + i * kWideBlock < e1) {
                    int u = c[i] - n0;
                    if ((unsigned)u >= (unsigned)ng) { fault |= DGCN_FAULT_BAD_COLUMN; u = ng; }
                    cl[base + i * kWideBlock - e0] = (uint16_t)u;
                }
        }
    }
    double wmax = 0.0;
    if (a.residual) {
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            cnt += __shfl_xor(cnt, off);
            pos |= __shfl_xor(pos, off);
            mx = fmax(mx, __shfl_xor(mx, off));
        }
        int* ri = reinterpret_cast<int*>(red + 16);
        if (lane == 0) { red[wave] = mx; ri[wave] = cnt; ri[16 + wave] = pos; }
        __syncthreads();
        cnt = 0; pos = 0; mx = -1.0 / 0.0;
#pragma unroll
        for (int w = 0; w < kWideBlock / 64; ++w) { cnt += ri[w]; pos |= ri[16 + w]; mx = fmax(mx, red[w]); }
        wmax = mx;
        // nothing left, or no positive weight left (np.sum(wts_nn) <= 0 -> break, mwis_gdpg_call.py:286): the graph is left alone
        if (!pos) {
            if (threadIdx.x == 0) {
                if (a.rounds) a.rounds[g] = 0;
                if (a.totals) a.totals[g] = 0.0;
                if (a.active) a.active[g] = 0;
            }
            if (!a.scores_given) for (int v = threadIdx.x; v < ng; v += kWideBlock) a.sc[n0 + v] = 0.f;
            if (fault) atomicOr(a.status, fault);
            return;
        }
    } else {
        __syncthreads();
    }
    // ---- scores.  The priorities' space holds d^-1/2 meanwhile, the row offsets' space z1 per vertex or the scores.
    const bool need_z1 = a.X != nullptr || a.feature_mode == 1 || a.two;
    float* z1 = need_z1 ? reinterpret_cast<float*>(rol) : nullptr;
    float* stash = need_z1 ? nullptr : reinterpret_cast<float*>(rol);
    if (!a.scores_given) {
        if (a.two) {  // (the layers' constants in the reduction array's space: free until the totals)
            if (cols_lds) wide_scores2<true>(a, n0, ng, e0, st, cl, pr, z1, reinterpret_cast<float*>(red + 32), fault);
            else wide_scores2<false>(a, n0, ng, e0, st, cl, pr, z1, reinterpret_cast<float*>(red + 32), fault);
        } else if (cols_lds) wide_scores<true>(a, n0, ng, e0, st, cl, pr, z1, stash, wmax, fault);
        else wide_scores<false>(a, n0, ng, e0, st, cl, pr, z1, stash, wmax, fault);
    } else {
        stash = nullptr;
    }
    if (fault) atomicOr(a.status, fault);
    __syncthreads();  // every chain has read its d^-1/2 and z1; the scores of this graph are visible to its workgroup
    // ---- priorities (mwis_dqn_call.py:230-235: float32 x float64 -> float64); a decided vertex: 0, takes part in nothing
    int bad = 0;
    double pmine[(kWideMaxNodes + kWideBlock - 1) / kWideBlock];
#pragma unroll
    for (int p = 0; p < (kWideMaxNodes + kWideBlock - 1) / kWideBlock; ++p) {
        const int v = threadIdx.x + p * kWideBlock;
        double q = 0.0;
        if (v < ng && st[v] == 0) {
            q = (double)(stash ? stash[v] : a.sc[n0 + v]);
            if (a.predict_mwis && a.weights) q *= a.weights[n0 + v];
            bad |= q != q;
        }
        pmine[p] = q;
    }
    __syncthreads();  // (the stash lies where the row offsets go)
#pragma unroll
    for (int p = 0; p < (kWideMaxNodes + kWideBlock - 1) / kWideBlock; ++p) {
        const int v = threadIdx.x + p * kWideBlock;
        if (v < ng) {
            pr[v] = pmine[p];
            if (a.prio) a.prio[n0 + v] = pmine[p];
        }
    }
    for (int v = threadIdx.x; v <= ng; v += kWideBlock) rol[v] = a.row_ptr[n0 + v];
    if (__syncthreads_or(bad)) {
        // the reference would spin forever on a NaN priority (its argmax would pick it): report, leave the graph as it is
        if (threadIdx.x == 0) {
            atomicOr(a.status, DGCN_FAULT_NAN_PRIORITY);
            if (a.rounds) a.rounds[g] = -1;
            if (a.totals) a.totals[g] = 0.0;
            if (a.active) a.active[g] = 0;
        }
        if (!a.residual) for (int v = threadIdx.x; v < ng; v += kWideBlock) a.state[n0 + v] = 0;
        return;
    }
    if (threadIdx.x == 0 && a.residual) {
        if (a.progress) atomicAdd(a.progress, 1);
        if (a.tail_word) atomicMax(a.tail_word, a.tail_tag | (unsigned long long)(unsigned)cnt);
    }
    if (a.mode == 2) {  // the rollout: candidates, completions and the pick right here, or general.hip's launches behind this one
        if (threadIdx.x == 0 && a.active) a.active[g] = 1;
        if (a.cid) {  // the candidates right here, the priorities still in registers: no k_res_cand launch
            static_assert(kCandPer * kWideBlock >= kWideMaxNodes && kCandPer == (kWideMaxNodes + kWideBlock - 1) / kWideBlock, "pmine is the selection's layout");
            unsigned have = 0u;
#pragma unroll
            for (int p = 0; p < kCandPer; ++p) {
                const int v = threadIdx.x + p * kWideBlock;
                if (v < ng && st[v] == 0) have |= 1u << p;
            }
            if (a.roll_off) {
                // the whole step here: candidates (list mirrored in LDS), the completions of all of them at once, the pick
                unsigned char* tail = wide_raw + a.roll_off;
                int32_t* cidl = reinterpret_cast<int32_t*>(tail + ((cand_scratch_bytes(kWideBlock) + 15) & ~(size_t)15));
                if (threadIdx.x < kCandMaxBeam) cidl[threadIdx.x] = -1;
                __syncthreads();
                cand_select<kWideBlock>(pmine, have, (ng + kWideBlock - 1) / kWideBlock, min(a.beam, kCandMaxBeam),
                                        a.cid + (size_t)g * kCandMaxBeam, tail, cidl);
                if (!a.by_priority)  // (the completions go by weight: mwis_gdpg_call.py:640)
                    for (int v = threadIdx.x; v < ng; v += kWideBlock) pr[v] = a.weights[n0 + v];
                __syncthreads();
                RollArgs r;
                r.ng = ng; r.n0 = n0; r.e0 = e0;
                r.key = pr; r.st = st; r.rol = rol; r.cl = cl; r.cidl = cidl; r.beam = a.beam;
                r.extra = reinterpret_cast<unsigned char*>(cidl + kCandMaxBeam);
                r.max_nodes = a.max_nodes;
                r.wl = a.by_priority ? nullptr : pr; r.weights = a.weights; r.state = a.state; r.rounds = a.rounds; r.totals = a.totals;
                rollout_bits<kWideBlock>(r, g);
                return;
            }
            __syncthreads();  // (the selection's scratch takes the carve from its start: nothing of it is read again)
            cand_select<kWideBlock>(pmine, have, (ng + kWideBlock - 1) / kWideBlock, min(a.beam, kCandMaxBeam),
                                    a.cid + (size_t)g * kCandMaxBeam, wide_raw);
        }
        return;
    }
    if (a.mode == 1) {
        // ---- solve_mwis_cit: the best-priority undecided vertex joins (np.argmax: lowest index among equals), its neighbours leave
        double bp = 0.0;
        int bv = -1;
        for (int v = threadIdx.x; v < ng; v += kWideBlock) {
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
        for (int w = 1; w < kWideBlock / 64; ++w) {
            const double op = red[w];
            const int ov = ri[w];
            if (ov >= 0 && (bv < 0 || op > bp || (op == bp && ov < bv))) { bp = op; bv = ov; }
        }
        if (bv < 0) return;  // (cannot happen: the graph has an undecided vertex)
        for (int j = rol[bv] + (int)threadIdx.x; j < rol[bv + 1]; j += kWideBlock) {
            const int u = a.col_idx[j] - n0;
            if ((unsigned)u < (unsigned)ng && u != bv && st[u] == 0) a.state[n0 + u] = 2;
        }
        if (threadIdx.x == 0) {
            a.state[n0 + bv] = 1;
            if (a.rounds) a.rounds[g] = 1;
            if (a.totals) a.totals[g] = a.weights ? a.weights[n0 + bv] : bp;
        }
        return;
    }
    // ---- the local greedy search (heuristics.py:77-116), k_lgs's rounds on the state already in LDS
    LgsArgs la = {};
    la.col_idx = a.col_idx;
    la.max_rounds = a.max_rounds;
    la.rounds = a.rounds;
    la.init_state = a.residual ? a.state : nullptr;
    if (cols_lds && a.max_rounds <= 0 && ng <= 4096 && a.ahead_rounds) {
        // (a whole search on graphs whose ahead counts fit the reduction array's space, free until the totals)
        const int rounds = lgs_rounds_ahead<kWideBlock, false>(ng, e0, pr, st, nw, cl, rol, reinterpret_cast<uint16_t*>(red), acc64);
        if (threadIdx.x == 0 && a.rounds) a.rounds[g] = rounds;
        __syncthreads();
    } else
    if (cols_lds) lgs_rounds<LPV, false, true, kWideBlock>(la, g, n0, ng, e0, pr, st, nw, cl, acc64, rol);
    else lgs_rounds<LPV, false, false, kWideBlock>(la, g, n0, ng, e0, pr, st, nw, cl, acc64, rol);
    {
        // state out; totals as k_lgs forms them: strided partials folded to 256 slots, then a binary tree.  A vertex that was a
        // member before this step is not counted (the total is what joined in THIS call).
        double part = 0.0;
        for (int v = threadIdx.x; v < ng; v += kWideBlock) {
            const uint8_t s1 = st[v];
            if (a.totals && s1 == 1 && !(a.residual && a.state[n0 + v] == 1)) part += a.weights ? a.weights[n0 + v] : pr[v];
            a.state[n0 + v] = s1;
        }
        red[threadIdx.x] = part;
    }
    if (a.totals) {
        __syncthreads();
        if (threadIdx.x < 256) {
            double acc = red[threadIdx.x];
            for (int k2 = 256; k2 < kWideBlock; k2 += 256) acc += red[threadIdx.x + k2];
            red[threadIdx.x] = acc;
        }
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) {
            if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.totals[g] = red[0];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// which (batch, model) pairs: one layer F -> 1 over [I, L], F <= 64, graphs of at most 9 600 vertices.  Option "wide1" = 0
// sends them layer by layer instead (tests compare the two).
int wide1_takes(const DgcnBatch* b, const DgcnModel* m, const float* X, int32_t feature_mode) {
    const bool off = opt(OPT_WIDE1) == 0;
    if (off || !b || !m || !m->layers_host || m->num_supports != 2) return 0;
    if (b->max_nodes <= 0 || b->max_nodes > kWideMaxNodes) return 0;
    const DgcnLayer& L = m->layers_host[0];
    if (!L.weights || L.in_dim < 1 || L.in_dim > 64) return 0;
    if (m->num_layers == 1) return L.out_dim == 1;
    // two layers F -> C -> 1, C <= 64, on constant input features (per-vertex features would need the neighbours' C-wide rows)
    const bool off2 = opt(OPT_WIDE2) == 0;
    if (m->num_layers != 2 || off2 || X || feature_mode != 0) return 0;
    const DgcnLayer& L1 = m->layers_host[1];
    return L.out_dim >= 1 && L.out_dim <= 64 && L1.weights && L1.in_dim == L.out_dim && L1.out_dim == 1;
}

template <int LPV>
static int wide1_launch_l(const WideArgs& a, int B, size_t lds, const char* family, hipStream_t s) {
    if (lds > 64 * 1024) {
        // (the dynamic limit is raised to what is asked for, not to the CU's 160 KB: the kernel's workgroup votes bring a few
        // bytes of static LDS with them, and static + dynamic must fit)
        static std::atomic<size_t> raised[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (raised[dev & 63].load(std::memory_order_relaxed) < lds) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide1<LPV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return fail(DGCN_ERR_LAUNCH, "k_wide1: cannot reserve %zu bytes of LDS", lds);
            raised[dev & 63].store(lds, std::memory_order_relaxed);
        }
    }
    TimedLaunch t(family, s);
    DGCN_LAUNCH(t, (k_wide1<LPV>), dim3((unsigned)B), dim3(kWideBlock), lds, s, a);
    return check_launch("k_wide1");
}

// One launch: dgcn_solve_batch (residual = 0) or one step of dgcn_solve_residual_batch (residual = 1) for a one-layer model.
// `sc`: the scores array in the original numbering (the caller's, or scratch); `prio` / `active`: mode 2 only.
int wide1_run(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, const float* X, float x_const,
              int32_t feature_mode, const double* weights, int32_t predict_mwis, int32_t residual, int32_t scores_given, int32_t mode,
              int32_t max_rounds, float* sc, uint8_t* state, int32_t* rounds, double* totals, int32_t* progress, int32_t* status,
              double* prio, int32_t* active, int32_t* cid, int32_t beam, int32_t by_priority, int32_t* whole_step,
              unsigned long long* tail_word, unsigned long long tail_tag, hipStream_t s) {
    const DgcnLayer& L = m->layers_host[0];
    if (whole_step) *whole_step = 0;
    WideArgs a = {};
    a.graph_ptr = b->graph_ptr; a.row_ptr = b->row_ptr; a.col_idx = b->col_idx;
    a.dinv = dinv_table; a.table_len = table_len;
    a.X = X; a.x_const = x_const; a.cin = L.in_dim; a.feature_mode = feature_mode;
    a.W = L.weights; a.bias = L.bias; a.act = L.act;
    if (m->num_layers == 2) {
        const DgcnLayer& L1 = m->layers_host[1];
        a.two = 1; a.C = L.out_dim; a.W2 = L1.weights; a.bias2 = L1.bias; a.act2 = L1.act;
    }
    a.weights = weights; a.predict_mwis = predict_mwis; a.residual = residual; a.scores_given = scores_given; a.mode = mode;
    a.max_rounds = max_rounds;
    a.sc = sc; a.state = state; a.rounds = rounds; a.totals = totals; a.status = status; a.progress = progress;
    a.prio = mode == 2 ? prio : nullptr; a.active = mode == 2 ? active : nullptr;
    a.cid = mode == 2 ? cid : nullptr; a.beam = beam;
    a.tail_word = tail_word; a.tail_tag = tail_tag;
    a.max_nodes = std::max(b->max_nodes, 16);
    // the graph's columns in LDS when the largest graph's fit next to the search's state
    constexpr size_t kLdsMax = 156 * 1024;
    int cap = std::max(b->max_graph_edges, 0);
    if (wide_lds_bytes(a.max_nodes, cap) > kLdsMax) cap = 0;
    a.cols_cap = cap;
    size_t lds = wide_lds_bytes(a.max_nodes, cap);
    if (a.cid) lds = std::max(lds, (size_t)cand_scratch_bytes(kWideBlock));  // (small graphs: the selection's scratch is the larger)
    a.by_priority = by_priority;
    a.ahead_rounds = opt(OPT_WIDE_AHEAD) == 0 ? 0 : 1;
    if (a.cid && whole_step) {
        // the completions and the pick in this launch too (rollout_bits.h) when sixteen candidates do and the columns, the
        // selection's scratch and the instances' state words all fit the LDS.  Option "rollout_bits" = 0: general.hip's launches.
        const bool bits_off = opt(OPT_ROLLOUT_BITS) == 0;
        const size_t base = (wide_lds_bytes(a.max_nodes, cap) + 15) & ~(size_t)15;
        const size_t need = base + ((cand_scratch_bytes(kWideBlock) + 15) & ~(size_t)15) + kCandMaxBeam * 4 + roll_lds_bytes(a.max_nodes);
        if (!bits_off && beam <= kRollBeam && weights && (cap > 0 || b->max_graph_edges == 0) && need <= kLdsMax) {
            a.roll_off = (int32_t)base;
            lds = need;
            *whole_step = 1;
        }
    }
    if (lds > 160 * 1024) return fail(DGCN_ERR_UNSUPPORTED, "k_wide1: %zu bytes of LDS for graphs of %d vertices", lds, b->max_nodes);
    const char* family = residual ? "wide_residual" : "wide_solve";
    // lanes per vertex in the rounds: k_lgs's choice for 1 024-thread workgroups
    const int lpv = b->max_nodes <= 256 ? 4 : (b->max_nodes <= 512 ? 2 : 1);
    if (lpv == 4) return wide1_launch_l<4>(a, b->num_graphs, lds, family, s);
    if (lpv == 2) return wide1_launch_l<2>(a, b->num_graphs, lds, family, s);
    return wide1_launch_l<1>(a, b->num_graphs, lds, family, s);
}

}  // namespace dgcn
