// Local greedy MWIS search over a batch of graphs: one workgroup per graph, state in LDS.
// Replaces heuristics.py:77-116 local_greedy_search and its instrumented twins (:119-305), and -
// because the lexicographically-first independent set under the order (priority desc, index asc)
// is unique - heuristics.py:13-35 greedy_search as well.  Also folds in the priority product
// gcn_wts = act_vals * wts (mwis_dqn_call.py:232; f32 x f64 -> f64).
//
// Per round (heuristics.py:90-114), synchronously for every vertex still remaining:
//   phase A  v wins iff no remaining neighbour u has (p[u] > p[v]) or (p[u] == p[v] and u < v)
//   phase B  winners join the set (state 1) and push state 2 ("nb_is") onto remaining neighbours
// Integer/compare work only: results are bit-identical to the reference for identical priorities.
// Signed priorities, +-0.0 and +-inf follow IEEE compares exactly as NumPy does; NaN (on which the
// reference never terminates) raises DGCN_FAULT_NAN_PRIORITY instead.
//
// gfx950 mapping: LPV lanes share one vertex and stride over its adjacency list; the per-vertex
// "lost" verdict is the OR over those lanes, taken from one wave-wide __ballot (64-bit mask, the
// lanes of a vertex are adjacent bits) instead of a shuffle tree.  Priorities (f64), state bytes and
// the graph's column ids (as 16-bit local ids) sit in LDS for all rounds, so HBM is touched once:
// nnz*4 + N*(8..12) bytes in, N*1 + 4 bytes out per graph.  Latency-bound at N~200, not HBM-bound.
#include "common.h"
#include "lgs_rounds.h"

namespace dgcn {

template <int LPV, bool STATS, int BLOCK = 256>
__global__ __launch_bounds__(BLOCK) void k_lgs(LgsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int inst = blockIdx.x / a.num_graphs;
    const int g = blockIdx.x - inst * a.num_graphs;
    if (inst) {
        const size_t no = (size_t)inst * a.num_nodes, go = (size_t)inst * a.num_graphs;
        a.state += no;
        if (a.init_state && !a.cand) a.init_state += no;
        if (a.overhead) a.overhead += no;
        if (a.rounds) a.rounds += go;
        if (a.totals) a.totals += go;
        if (a.stats) a.stats += 2 * go;
        if (a.prio) a.prio += (size_t)inst * a.prio_stride;
    }
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1];
    const int cnd = a.cand ? a.cand[(size_t)g * 64 + inst] : 0;  // (fewer undecided vertices than candidates: nothing to search)
    const int ng = ((a.active && !a.active[g]) || cnd < 0) ? 0 : n1 - n0;
    // carve: [f64 prio | f64 reduce[256] | u64 acc[4] | i32 row offsets | u8 st | u8 nw | u16 cols]
    double* pr = reinterpret_cast<double*>(lds_raw);
    double* red = pr + a.max_nodes;
    unsigned long long* acc64 = reinterpret_cast<unsigned long long*>(red + 1024);
    int* rol = reinterpret_cast<int*>(acc64 + 4);
    uint8_t* st = reinterpret_cast<uint8_t*>(rol + ((a.max_nodes + 1 + 3) & ~3));
    uint8_t* nw = st + ((a.max_nodes + 15) & ~15);
    uint16_t* cl = reinterpret_cast<uint16_t*>(nw + ((a.max_nodes + 15) & ~15));
    if (ng <= 0) {
        if (threadIdx.x == 0) {
            if (a.rounds) a.rounds[g] = 0;
            if (a.stats) { a.stats[2 * g] = 0; a.stats[2 * g + 1] = 0; }
            if (a.totals) a.totals[g] = 0.0;
        }
        return;
    }
    int bad = 0;
    for (int v = threadIdx.x; v < ng; v += BLOCK) {
        double p;
        if (a.prio) p = a.prio[n0 + v];
        else if (a.weights) p = (double)a.scores[n0 + v] * a.weights[n0 + v];
        else p = (double)a.scores[n0 + v];
        const uint8_t s0 = a.init_state ? a.init_state[n0 + v] : (uint8_t)0;
        bad |= (p != p) && s0 == 0;  // a masked-out vertex may carry any priority
        pr[v] = p;
        st[v] = s0;
        nw[v] = 0;
        if (STATS && a.overhead) a.overhead[n0 + v] = 0;
    }
    const int e0 = a.row_ptr[n0], e1 = a.row_ptr[n1];
    const bool cols_lds = (e1 - e0) <= a.cols_cap && ng <= 65536;
    for (int v = threadIdx.x; v <= ng; v += BLOCK) rol[v] = a.row_ptr[n0 + v];  // row bounds: read once, not per round
    if (cols_lds) {
        for (int base = e0 + threadIdx.x; base < e1; base += BLOCK * 4) {  // 4 loads in flight per thread
            int c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = (base + i * BLOCK < e1) ? a.col_idx[base + i * BLOCK] : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (base + i * BLOCK < e1) cl[base + i * BLOCK - e0] = (uint16_t)(c[i] - n0);
        }
    }
    if (a.cand) {  // this instance's start: the candidate and its neighbours do not take part
        __syncthreads();
        const int rs = a.row_ptr[n0 + cnd], re = a.row_ptr[n0 + cnd + 1];
        for (int j = rs + threadIdx.x; j < re; j += BLOCK) {
            const int u = a.col_idx[j] - n0;
            if ((unsigned)u < (unsigned)ng) st[u] = 3;
        }
        if (threadIdx.x == 0) st[cnd] = 3;
    }
    if (__syncthreads_or(bad)) {
        // the reference would spin forever on a NaN priority: report instead
        if (threadIdx.x == 0) {
            atomicOr(a.status, DGCN_FAULT_NAN_PRIORITY);
            if (a.rounds) a.rounds[g] = -1;
            if (a.stats) { a.stats[2 * g] = 0; a.stats[2 * g + 1] = 0; }
            if (a.totals) a.totals[g] = 0.0;
        }
        for (int v = threadIdx.x; v < ng; v += BLOCK) a.state[n0 + v] = a.init_state ? a.init_state[n0 + v] : (uint8_t)0;
        return;
    }
    if (!STATS && cols_lds && a.max_rounds <= 0 && ng <= 4096 && a.ahead_rounds) {
        // a whole search without statistics: the rounds on ahead lists (lgs_rounds.h: two walks over state / flag bytes per round;
        // the 16-bit counts in the reduction array's space, free until the totals)
        const int rounds = lgs_rounds_ahead<BLOCK, false>(ng, e0, pr, st, nw, cl, rol, reinterpret_cast<uint16_t*>(red), acc64);
        if (threadIdx.x == 0 && a.rounds) a.rounds[g] = rounds;
        __syncthreads();
    } else
    if (cols_lds) lgs_rounds<LPV, STATS, true, BLOCK>(a, g, n0, ng, e0, pr, st, nw, cl, acc64, rol);
    else lgs_rounds<LPV, STATS, false, BLOCK>(a, g, n0, ng, e0, pr, st, nw, cl, acc64, rol);

    {
        // totals: fixed-shape reduction - strided partials, then a binary tree over the 256 slots.  A vertex that was given
        // as a member already (init_state 1: the residual steps of general.hip pass the running state) is not counted:
        // the total is what joined in THIS search.  (init_state may alias state: read before the write, same thread.)
        const double* sw = a.sum_weights;
        double part = 0.0;
        for (int v = threadIdx.x; v < ng; v += BLOCK) {
            const uint8_t s1 = st[v];
            if (a.totals && s1 == 1 && !(a.init_state && a.init_state[n0 + v] == 1)) part += sw ? sw[n0 + v] : pr[v];
            a.state[n0 + v] = s1;
        }
        red[threadIdx.x] = part;
    }
    if (a.totals) {
        __syncthreads();
        // (the partial of thread i covers vertices i, i + BLOCK, ..; folded to 256 slots - slot j = partials j, j + 256, .. added
        // in order - and then the same tree: a fixed shape per block size, float64 sums equal to 1e-16 relative either way)
        if (BLOCK > 256) {
            if (threadIdx.x < 256) {
                double acc = red[threadIdx.x];
                for (int k2 = 256; k2 < BLOCK; k2 += 256) acc += red[threadIdx.x + k2];
                red[threadIdx.x] = acc;
            }
            __syncthreads();
        }
        for (int off = 128; off > 0; off >>= 1) {
            if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.totals[g] = red[0];
    }
}

static size_t lgs_lds_bytes(int max_nodes, int cols_cap) {
    const size_t pad = (size_t)((max_nodes + 15) & ~15);
    const size_t ro = (size_t)((max_nodes + 1 + 3) & ~3) * 4;
    return (size_t)max_nodes * 8 + 1024 * 8 + 4 * 8 + ro + 2 * pad + (size_t)cols_cap * 2;
}

template <int LPV, bool STATS, int BLOCK = 256>
static int launch_lgs(const LgsArgs& a, int B, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lgs<LPV, STATS, BLOCK>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DGCN_ERR_LAUNCH, "k_lgs: cannot reserve %zu bytes of LDS", lds);
    }
    TimedLaunch t("lgs", s);
    DGCN_LAUNCH(t, (k_lgs<LPV, STATS, BLOCK>), dim3(B), dim3(BLOCK), lds, s, a);
    return check_launch("k_lgs");
}

// ---------------------------------------------------------------------------------------------
// which selected sets could a score error flip?
// The local greedy search returns the lexicographically-first maximal independent set S under the total order
// (priority desc, index asc), and S is THE set with: S independent, and every vertex outside S has a neighbour
// in S that precedes it.  So S survives any perturbation of the priorities that keeps, for every excluded
// vertex v, at least one member neighbour u ahead of it.  With every score off by at most `delta`, priorities
// p = score * w move by at most delta * |w|, hence v is SAFE iff some member neighbour u has
//     p_u - p_v > delta * (|w_u| + |w_v|)
// (a tie or a thinner margin is not safe).  risky[g] = number of excluded vertices of graph g that are not safe;
// risky[g] == 0 proves that graph's set is the same for every score vector within delta of the given one.
// 8 lanes per vertex, one pass over the adjacency; runs after the search, outside the solver launch.
__global__ __launch_bounds__(256) void k_margin_risk(const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ row_ptr,
                                                     const int32_t* __restrict__ col_idx, const double* __restrict__ prio,
                                                     const float* __restrict__ scores, const double* __restrict__ weights,
                                                     const uint8_t* __restrict__ state, double delta,
                                                     int32_t* __restrict__ risky) {
    const int g = blockIdx.x;
    const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1];
    const int sub = threadIdx.x & 7;
    auto pri = [&](int v) -> double {
        if (prio) return prio[v];
        return weights ? (double)scores[v] * weights[v] : (double)scores[v];
    };
    auto wabs = [&](int v) -> double { return (weights && !prio) ? fabs(weights[v]) : 1.0; };
    int count = 0;
    for (int v0 = n0 + (threadIdx.x >> 3); v0 - (int)(threadIdx.x >> 3) < n1; v0 += 32) {
        const int v = v0;
        bool need = v < n1 && state[v] == 2, safe = false;
        if (need) {
            const double pv = pri(v), wv = wabs(v);
            for (int j = row_ptr[v] + sub; j < row_ptr[v + 1]; j += 8) {
                const int u = col_idx[j];
                if (state[u] == 1) safe |= (pri(u) - pv) > delta * (wabs(u) + wv);
            }
        }
        safe |= __shfl_xor((int)safe, 1) != 0;
        safe |= __shfl_xor((int)safe, 2) != 0;
        safe |= __shfl_xor((int)safe, 4) != 0;
        count += (need && !safe && sub == 0);
    }
    __shared__ int total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    if (count) atomicAdd(&total, count);
    __syncthreads();
    if (threadIdx.x == 0) risky[g] = total;
}

}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_margin_risk_batch(const DgcnBatch* b, const double* prio, const float* scores, const double* weights,
                                      const uint8_t* state, double delta, int32_t* risky, void* stream) {
    if (!b || !state || !risky || (!prio && !scores)) return fail(DGCN_ERR_ARG, "dgcn_margin_risk_batch: null argument");
    if (!(delta >= 0.0)) return fail(DGCN_ERR_ARG, "dgcn_margin_risk_batch: delta must be >= 0");
    if (b->num_graphs <= 0) return DGCN_OK;
    TimedLaunch t("margin_risk", (hipStream_t)stream);
    DGCN_LAUNCH(t, k_margin_risk, dim3(b->num_graphs), dim3(256), 0, (hipStream_t)stream, b->graph_ptr, b->row_ptr, b->col_idx,
                prio, scores, weights, state, delta, risky);
    return check_launch("k_margin_risk");
}

namespace dgcn {
int lgs_launch_common(const DgcnBatch* b, const double* prio, long prio_stride, const float* scores,
                      const double* weights, const uint8_t* init_state, int32_t num_instances,
                      int32_t max_rounds, uint8_t* state, int32_t* rounds, int64_t* stats, int32_t* overhead,
                      const double* sum_weights, double* totals, int32_t* status, void* stream, const int32_t* active = nullptr,
                      const int32_t* cand = nullptr);
}

extern "C" int dgcn_lgs_batch(const DgcnBatch* b, const double* prio, const float* scores, const double* weights,
                              int32_t max_rounds, uint8_t* state, int32_t* rounds, int64_t* stats, int32_t* overhead,
                              const double* sum_weights, double* totals, int32_t* status, void* stream) {
    return lgs_launch_common(b, prio, 0, scores, weights, nullptr, 1, max_rounds, state, rounds, stats, overhead,
                             sum_weights, totals, status, stream);
}

extern "C" int dgcn_lgs_masked_batch(const DgcnBatch* b, const double* prio, int64_t prio_stride,
                                     const uint8_t* init_state, int32_t num_instances, int32_t max_rounds,
                                     uint8_t* state, int32_t* rounds, const double* sum_weights, double* totals,
                                     int32_t* status, void* stream) {
    if (!prio || !init_state || num_instances <= 0)
        return fail(DGCN_ERR_ARG, "dgcn_lgs_masked_batch: prio, init_state and num_instances > 0 are required");
    return lgs_launch_common(b, prio, (long)prio_stride, nullptr, nullptr, init_state, num_instances, max_rounds, state,
                             rounds, nullptr, nullptr, sum_weights, totals, status, stream);
}

int dgcn::lgs_launch_common(const DgcnBatch* b, const double* prio, long prio_stride, const float* scores,
                            const double* weights, const uint8_t* init_state, int32_t num_instances,
                            int32_t max_rounds, uint8_t* state, int32_t* rounds, int64_t* stats, int32_t* overhead,
                            const double* sum_weights, double* totals, int32_t* status, void* stream, const int32_t* active,
                            const int32_t* cand) {
    if (!b || !state || !status) return fail(DGCN_ERR_ARG, "dgcn_lgs_batch: null argument");
    if (!prio && !scores) return fail(DGCN_ERR_ARG, "dgcn_lgs_batch: need prio or scores");
    if (b->num_graphs <= 0) return DGCN_OK;
    if (b->max_nodes <= 0) return fail(DGCN_ERR_ARG, "dgcn_lgs_batch: max_nodes must be positive");
    constexpr size_t kLdsMax = 156 * 1024;
    if (lgs_lds_bytes(b->max_nodes, 0) > kLdsMax)
        return fail(DGCN_ERR_UNSUPPORTED, "dgcn_lgs_batch: graphs of %d vertices exceed the per-workgroup LDS state",
                    b->max_nodes);
    LgsArgs a;
    a.graph_ptr = b->graph_ptr; a.row_ptr = b->row_ptr; a.col_idx = b->col_idx;
    a.prio = prio; a.scores = scores; a.weights = weights; a.max_rounds = max_rounds;
    a.state = state; a.rounds = rounds; a.stats = stats; a.overhead = overhead;
    a.sum_weights = sum_weights; a.totals = totals; a.status = status;
    a.max_nodes = b->max_nodes;
    a.init_state = init_state;
    a.num_graphs = b->num_graphs;
    a.num_nodes = b->num_nodes;
    a.prio_stride = prio_stride;
    a.active = active;
    a.cand = cand;
    a.ahead_rounds = opt(OPT_WIDE_AHEAD) == 0 ? 0 : 1;
    // column ids in LDS when the largest graph's adjacency fits next to the state (prefer <= 48 KB
    // per workgroup so several graphs share a CU; allow up to the whole LDS for big graphs)
    int cap = b->max_graph_edges > 0 ? b->max_graph_edges : 0;
    if (cap > 0 && lgs_lds_bytes(b->max_nodes, cap) > kLdsMax) cap = 0;
    a.cols_cap = cap;
    const size_t lds = lgs_lds_bytes(b->max_nodes, cap);
    hipStream_t s = (hipStream_t)stream;
    const bool want_stats = stats != nullptr || overhead != nullptr;
    const int lpv_env = opt(OPT_LGS_LPV);  // tuning / test knob
    int lpv = lpv_env > 0 ? lpv_env : (b->max_nodes <= 512 ? 4 : 1);  // measured: 4 lanes per vertex wins at N ~ 200
    // graphs beyond the fused kernel's sizes (the any-size path, general.hip): 1 024 threads per graph - a vertex's lanes in
    // every pass instead of a quarter of the graph per pass (ER(500, 0.1), 256 graphs: 118 us with 256 threads; the search is a chain
    // of LDS round trips, more waves hide more of them).  Same decisions, same totals (the reduction tree is fixed).
    const int blk_env = opt(OPT_LGS_BLOCK);
    // (the 1 024-thread kernels are built without the statistics: a caller that wants them never gets the wide launch)
    const bool wide = !want_stats && (blk_env >= 0 ? blk_env == 1024 : b->max_nodes > 384);
    if (wide && lpv_env <= 0) lpv = b->max_nodes <= 256 ? 4 : (b->max_nodes <= 512 ? 2 : 1);
    if (wide) {
        if (lpv == 1) return launch_lgs<1, false, 1024>(a, b->num_graphs * num_instances, lds, s);
        if (lpv == 2) return launch_lgs<2, false, 1024>(a, b->num_graphs * num_instances, lds, s);
        if (lpv == 4) return launch_lgs<4, false, 1024>(a, b->num_graphs * num_instances, lds, s);
    }
#define DGCN_LGS_CASE(L)                                                      \
    if (lpv == L) return want_stats ? launch_lgs<L, true>(a, b->num_graphs * num_instances, lds, s) \
                                    : launch_lgs<L, false>(a, b->num_graphs * num_instances, lds, s)
    DGCN_LGS_CASE(1);
    DGCN_LGS_CASE(2);
    DGCN_LGS_CASE(4);
    DGCN_LGS_CASE(8);
#undef DGCN_LGS_CASE
    return fail(DGCN_ERR_ARG, "dgcn_lgs_batch: DGCN_LGS_LPV must be 1, 2, 4 or 8");
}
