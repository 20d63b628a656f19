// The solvers for graphs of ANY size: what dgcn_solve_batch / dgcn_solve_residual_batch run when a graph's image does not
// fit one CU's LDS (fused.hip: <= 512 vertices and an entry list bounded by the 160 KB), entirely on the device.
//
// The reference runs every solver on whatever conflict graph it is given - the multi-channel scripts on the joint graph
// of K * nflows vertices (wireless_dqn_test_mc.py:161, 244-289; built by wireless_rollout_test_flood.py:98-133) - and
// re-slices the SciPy matrix to the undecided vertices before every GCN pass of an iterative solver
// (mwis_gdpg_call.py:284-285, 349-350, 605-612).  Here one call of dgcn_solve_residual_batch is one solver step for every
// graph of the batch, as in fused.hip, made of these launches (all on the caller's stream, nothing read back):
//
//   k_res_count    per graph: which vertices are undecided, is anything left to do (np.sum(wts_nn) <= 0 -> break), the
//                  largest undecided weight (feature_mode 1), and every undecided vertex's degree IN THE RESIDUAL GRAPH
//   k_res_scan     exclusive scans of the graphs' vertex / entry counts: where each residual graph starts in the compact batch
//   k_res_fill     the re-sliced batch itself: undecided vertices renumbered in index order (so every "lower index first"
//                  rule holds), the support L = I - D^-1/2 A D^-1/2 of the residual graph with the diagonal entry first
//                  (supports.hip's layout and float64 -> float32 expression), the features
//   (forward.hip)  the layer-by-layer forward pass on that batch: the same kernels, hence the same bits, as mode 0
//   k_res_scatter  scores back to the original numbering (0 for decided vertices), priorities score * weight in float64
//   then one of    k_lgs with the running state as its mask            (greedy_mode 0: solve_mwis_dit, :278-318)
//                  k_res_central                                       (greedy_mode 1: solve_mwis_cit, :343-384)
//                  k_res_cand -> k_lgs (beam instances, each masking its candidate's closed neighbourhood) -> k_res_pick
//                                                                      (greedy_mode 2: the rollout, :596-659; behind k_big / k_big2 /
//                                                                      k_wide1 the candidates come out of that launch: cand_select.h)
//
// Every step is integer / compare work or an expression that also appears in fused.hip, so states and scores equal the
// fused kernel's bit for bit on shapes both take (tests/test_gpu_general.py forces this path with dgcn_set_general(1)).
// HBM-bound throughout: the compaction reads the adjacency twice (count, fill) and writes the support once.
#include <atomic>

#include "common.h"
#include "cand_select.h"

namespace dgcn {

int layered_forward(const DgcnBatch* b, const DgcnCsr* const* sup, const DgcnModel* m, const float* X, float x_const,
                    float* scores, void* workspace, hipStream_t s);  // forward.hip
int lgs_launch_common(const DgcnBatch* b, const double* prio, long prio_stride, const float* scores, const double* weights,
                      const uint8_t* init_state, int32_t num_instances, int32_t max_rounds, uint8_t* state, int32_t* rounds,
                      int64_t* stats, int32_t* overhead, const double* sum_weights, double* totals, int32_t* status, void* stream,
                      const int32_t* active, const int32_t* cand);  // lgs.hip

// big.hip: the hidden stack of a deep c32 model in one launch (graphs up to 976 vertices); same bits as layered_forward
int big_takes(const DgcnBatch* b, const DgcnModel* m);
size_t big_workspace(const DgcnBatch* b, const DgcnModel* m);
int big_forward(const DgcnBatch* b, const DgcnCsr* lap, const DgcnModel* m, const float* X, float x_const, float* scores,
                void* lws, void* bws, int32_t* status, hipStream_t s);
int big_solve_takes(const DgcnBatch* b, const DgcnModel* m, const float* X);
int big_solve(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, float x_const, const double* weights,
              int32_t predict_mwis, float* scores, uint8_t* state, int32_t* rounds, double* totals, int32_t* status, void* bws,
              hipStream_t s);

int big_residual_takes(const DgcnBatch* b, const DgcnModel* m, const float* X, int32_t feature_mode, int32_t options);
int big_residual(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, float x_const, const double* weights,
                 int32_t predict_mwis, int32_t greedy_mode, int32_t max_rounds, float* scores, uint8_t* state, int32_t* rounds,
                 double* totals, int32_t* progress, int32_t* status, double* prio, int32_t* active, int32_t* cid, int32_t beam,
                 int32_t by_priority, int32_t* whole_step, unsigned long long* tail_word, unsigned long long tail_tag, void* bws,
                 hipStream_t s);

// big2.hip: the same for graphs of 977 .. 1 920 vertices (Z1 a feature half at a time)
int big2_takes(const DgcnBatch* b, const DgcnModel* m);
size_t big2_workspace(const DgcnBatch* b, const DgcnModel* m);
int big2_forward(const DgcnBatch* b, const DgcnCsr* lap, const DgcnModel* m, const float* X, float x_const, float* scores,
                 void* lws, void* bws, int32_t* status, hipStream_t s);
int big2_solve_takes(const DgcnBatch* b, const DgcnModel* m, const float* X);
int big2_solve(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, float x_const, const double* weights,
               int32_t predict_mwis, float* scores, uint8_t* state, int32_t* rounds, double* totals, int32_t* status, void* bws,
               hipStream_t s);

int big2_residual_takes(const DgcnBatch* b, const DgcnModel* m, const float* X, int32_t feature_mode, int32_t options);
int big2_residual(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, float x_const, const double* weights,
                  int32_t predict_mwis, int32_t greedy_mode, int32_t max_rounds, float* scores, uint8_t* state, int32_t* rounds,
                  double* totals, int32_t* progress, int32_t* status, double* prio, int32_t* active, int32_t* cid, int32_t beam,
                  int32_t by_priority, int32_t* whole_step, unsigned long long* tail_word, unsigned long long tail_tag, void* bws,
                  hipStream_t s);

// wide.hip: one-layer models on graphs of any size - the plain solve, or the score / priority / greedy part of a residual step,
// in one launch
int wide1_takes(const DgcnBatch* b, const DgcnModel* m, const float* X, int32_t feature_mode);
int wide1_run(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, const float* X, float x_const,
              int32_t feature_mode, const double* weights, int32_t predict_mwis, int32_t residual, int32_t scores_given, int32_t mode,
              int32_t max_rounds, float* sc, uint8_t* state, int32_t* rounds, double* totals, int32_t* progress, int32_t* status,
              double* prio, int32_t* active, int32_t* cid, int32_t beam, int32_t by_priority, int32_t* whole_step,
              unsigned long long* tail_word, unsigned long long tail_tag, hipStream_t s);

constexpr int kResBlock = 1024;  // (graphs of this path are large and batches of them small: 64 graphs x 256 threads left the chip idle)
constexpr int kMaxBeam = 64;

struct ResArgs {
    const int32_t* graph_ptr;
    const int32_t* row_ptr;
    const int32_t* col_idx;
    uint8_t* state;            // in / out: 0 undecided, 1 member, 2 excluded
    const double* weights;     // or null (every weight counts as 1 for "anything left", as 0 for totals: fused.hip)
    const double* dinv;
    int32_t table_len;
    int32_t feature_mode, cin;
    const float* X;            // [num_nodes][cin] or null
    int32_t predict_mwis, scores_given, beam, by_priority;
    int32_t num_graphs, num_nodes;
    // per graph
    int32_t* na;               // undecided vertices (0 for a graph that is left alone)
    int32_t* ne;               // adjacency entries between undecided vertices
    int32_t* active;
    double* wmax;
    int32_t* gptr2;            // [B + 1] graph_ptr of the compact batch
    int32_t* eoff;             // [B + 1] adjacency entries before each graph of the compact batch
    // per vertex
    int32_t* adeg;             // degree in the residual graph (undecided vertices of active graphs only)
    int32_t* cidx;             // original global id -> compact global id, -1 = not part of the residual batch
    int32_t* lrow;             // [num_nodes + 1] the compact batch's support: row pointers (rows past the last one: empty)
    int32_t* lcol;
    float* lval;
    float* Xc;                 // [.][cin] features of the compact batch, or null
    const float* sc;           // scores of the compact batch (forward output)
    float* scores;             // caller's array (original numbering): output, or input with scores_given
    double* prio;              // [num_nodes] priorities, original numbering
    int32_t* cid;              // [B][64] rollout candidates (local vertex ids), -1 = none
    const double* inst_totals; // [beam][B]
    int32_t* rounds;
    double* totals;
    int32_t* status;
    int32_t* progress;
    unsigned long long* tail_word;  // or null: atomicMax(tail_tag | undecided vertices of an active graph), read by tail.hip
    unsigned long long tail_tag;
};

// exclusive scan of two ints over the workgroup; returns the block totals through ta / tb.  `sh` = 2 * (BLOCK / 64) + 2 ints.
template <int BLOCK>
__device__ __forceinline__ void block_scan2(int a, int b, int& xa, int& xb, int& ta, int& tb, int* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int ia = a, ib = b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int pa = __shfl_up(ia, off), pb = __shfl_up(ib, off);
        if (lane >= off) { ia += pa; ib += pb; }
    }
    __syncthreads();  // (sh may still be read from the previous call)
    if (lane == 63) { sh[2 * wave] = ia; sh[2 * wave + 1] = ib; }
    __syncthreads();
    int ba = 0, bb = 0, sa = 0, sb = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        const int wa = sh[2 * w], wb = sh[2 * w + 1];
        if (w < wave) { ba += wa; bb += wb; }
        sa += wa;
        sb += wb;
    }
    xa = ba + ia - a;
    xb = bb + ib - b;
    ta = sa;
    tb = sb;
}

// ---- 1. per graph: undecided vertices, "anything left?", largest undecided weight, residual degrees
__global__ __launch_bounds__(kResBlock) void k_res_count(ResArgs a) {
    __shared__ int s_cnt[kResBlock / 64], s_pos[kResBlock / 64], s_ne;
    __shared__ double s_max[kResBlock / 64];
    const int g = blockIdx.x;
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1], ng = n1 - n0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cnt = 0, pos = 0;
    double mx = -1.0 / 0.0;
    for (int v = threadIdx.x; v < ng; v += kResBlock) {
        const bool alive = a.state[n0 + v] == 0;
        const double w = a.weights ? a.weights[n0 + v] : 1.0;
        cnt += alive;
        pos |= alive && w > 0.0;
        if (alive) mx = fmax(mx, w);
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        cnt += __shfl_xor(cnt, off);
        pos |= __shfl_xor(pos, off);
        mx = fmax(mx, __shfl_xor(mx, off));
    }
    if (lane == 0) { s_cnt[wave] = cnt; s_pos[wave] = pos; s_max[wave] = mx; }
    if (threadIdx.x == 0) s_ne = 0;
    __syncthreads();
    cnt = 0; pos = 0; mx = -1.0 / 0.0;
#pragma unroll
    for (int w = 0; w < kResBlock / 64; ++w) { cnt += s_cnt[w]; pos |= s_pos[w]; mx = fmax(mx, s_max[w]); }
    // nothing left, or no positive weight left (np.sum(wts_nn) <= 0 -> break, mwis_gdpg_call.py:286): the graph is left alone
    if (!pos) {
        if (threadIdx.x == 0) { a.na[g] = 0; a.ne[g] = 0; a.active[g] = 0; a.wmax[g] = 0.0; }
        return;
    }
    int fault = 0, ne = 0;
    if (!a.scores_given) {
        // residual degrees: 8 lanes per row, 32 rows at a time
        const int sub = threadIdx.x & 7;
        for (int v0 = 0; v0 < ng; v0 += kResBlock / 8) {
            const int v = v0 + (threadIdx.x >> 3);
            const bool alive = v < ng && a.state[n0 + v] == 0;
            int c = 0;
            if (alive) {
                const int rs = a.row_ptr[n0 + v], re = a.row_ptr[n0 + v + 1];
                for (int j = rs + sub; j < re; j += 8) {
                    const int u = a.col_idx[j] - n0;
                    if (u < 0 || u >= ng) { fault |= DGCN_FAULT_BAD_COLUMN; continue; }
                    if (u == v) fault |= DGCN_FAULT_SELF_LOOP;
                    c += a.state[n0 + u] == 0;
                }
            }
            c += __shfl_xor(c, 1);
            c += __shfl_xor(c, 2);
            c += __shfl_xor(c, 4);
            if (alive && sub == 0) {
                a.adeg[n0 + v] = c;
                ne += c;
                if (c >= a.table_len) fault |= DGCN_FAULT_DEGREE_RANGE;
            }
        }
        if (ne) atomicAdd(&s_ne, ne);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.na[g] = cnt;
        a.ne[g] = s_ne;
        a.active[g] = 1;
        a.wmax[g] = mx;
        if (a.progress) atomicAdd(a.progress, 1);
        if (a.tail_word) atomicMax(a.tail_word, a.tail_tag | (unsigned long long)(unsigned)cnt);
    }
    if (fault) atomicOr(a.status, fault);
}

// ---- 2. where every residual graph starts in the compact batch
__global__ __launch_bounds__(1024) void k_res_scan(ResArgs a) {
    __shared__ int sh[2 * 16 + 2];
    int carry_n = 0, carry_e = 0;
    for (int base = 0; base < a.num_graphs; base += 1024) {
        const int i = base + threadIdx.x;
        const int n = i < a.num_graphs ? a.na[i] : 0, e = i < a.num_graphs ? a.ne[i] : 0;
        int xn, xe, tn, te;
        block_scan2<1024>(n, e, xn, xe, tn, te, sh);
        if (i < a.num_graphs) { a.gptr2[i] = carry_n + xn; a.eoff[i] = carry_e + xe; }
        carry_n += tn;
        carry_e += te;
    }
    if (threadIdx.x == 0) { a.gptr2[a.num_graphs] = carry_n; a.eoff[a.num_graphs] = carry_e; }
}

// ---- 3. the re-sliced batch: renumbering, support of the residual graph (diagonal first), features
__global__ __launch_bounds__(kResBlock) void k_res_fill(ResArgs a) {
    __shared__ int sh[2 * (kResBlock / 64) + 2];
    const int g = blockIdx.x, B = a.num_graphs;
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1], ng = n1 - n0;
    const int c0 = a.gptr2[g], e0 = a.eoff[g], nag = a.gptr2[g + 1] - c0;
    const int na_total = a.gptr2[B], end_all = a.eoff[B] + na_total;  // entries of the whole support
    // rows of the compact batch past its last vertex are empty (the row-parallel kernels walk all num_nodes rows): this
    // graph fills as many of them as it has vertices that are not part of the residual batch
    const int tail0 = na_total + (n0 - c0);
    for (int t = threadIdx.x; t < ng - nag; t += kResBlock) a.lrow[tail0 + t] = end_all;
    if (g == B - 1 && threadIdx.x == 0) a.lrow[a.num_nodes] = end_all;
    if (!a.active[g]) {
        for (int v = threadIdx.x; v < ng; v += kResBlock) a.cidx[n0 + v] = -1;
        return;
    }
    const double wmax = a.wmax[g];
    int carry_c = 0, carry_e = 0, fault = 0;
    for (int base = 0; base < ng; base += kResBlock) {
        const int v = base + threadIdx.x;
        const bool alive = v < ng && a.state[n0 + v] == 0;
        const int d = alive ? a.adeg[n0 + v] : 0;
        int xc, xe, tc, te;
        block_scan2<kResBlock>(alive ? 1 : 0, d, xc, xe, tc, te, sh);
        if (v < ng) {
            if (alive) {
                const int c = c0 + carry_c + xc;
                a.cidx[n0 + v] = c;
                a.lrow[c] = e0 + carry_e + xe + c;  // one diagonal entry per earlier row
                if (a.Xc) {
                    if (a.feature_mode == 1) {
                        // mwis_gdpg_call.py:88: wts_nn / (np.amax(wts_nn) + 1e-9), cast to float32 at the feed
                        const float f = (float)((a.weights ? a.weights[n0 + v] : 1.0) / (wmax + 1e-9));
                        for (int k = 0; k < a.cin; ++k) a.Xc[(size_t)c * a.cin + k] = f;
                    } else {
                        for (int k = 0; k < a.cin; ++k) a.Xc[(size_t)c * a.cin + k] = a.X[(size_t)(n0 + v) * a.cin + k];
                    }
                }
            } else {
                a.cidx[n0 + v] = -1;
            }
        }
        carry_c += tc;
        carry_e += te;
    }
    __syncthreads();  // cidx / lrow of the whole graph are written (same workgroup: visible after the barrier)
    const int sub = threadIdx.x & 7, grp = (threadIdx.x & 63) >> 3;
    for (int v0 = 0; v0 < ng; v0 += kResBlock / 8) {
        const int v = v0 + (threadIdx.x >> 3);
        const bool alive = v < ng && a.state[n0 + v] == 0;
        int rs = 0, re = 0, out = 0;
        double dv = 0.0;
        if (alive) {
            rs = a.row_ptr[n0 + v];
            re = a.row_ptr[n0 + v + 1];
            const int c = a.cidx[n0 + v];
            out = a.lrow[c];
            const int deg = a.adeg[n0 + v];
            if (deg < a.table_len) dv = a.dinv[deg]; else fault |= DGCN_FAULT_DEGREE_RANGE;
            if (sub == 0) { a.lcol[out] = c; a.lval[out] = 1.0f; }  // (I - A_hat)[v][v], zero-diagonal adjacency
        }
        int base = 1;
        for (int j0 = rs; __any(j0 < re); j0 += 8) {  // (every lane of the wave takes part in the ballot)
            const int j = j0 + sub;
            int u = -1;
            if (j < re) {
                u = a.col_idx[j] - n0;
                if (u < 0 || u >= ng || a.state[n0 + u] != 0) u = -1;
            }
            const bool keep = u >= 0;
            const unsigned bits = (unsigned)(__ballot(keep) >> (grp * 8)) & 0xffu;
            if (keep) {
                const int du = a.adeg[n0 + u];
                double d = 0.0;
                if (du < a.table_len) d = a.dinv[du]; else fault |= DGCN_FAULT_DEGREE_RANGE;
                const int slot = out + base + __popc(bits & ((1u << sub) - 1u));
                a.lcol[slot] = a.cidx[n0 + u];
                // reference order: (A_vu * dinv[u]) * dinv[v] in float64, negated by "eye - A_hat", then the float32 feed cast
                a.lval[slot] = (float)(-(d * dv));
            }
            base += __popc(bits);
        }
    }
    if (fault) atomicOr(a.status, fault);
}

// ---- 4. scores back to the original numbering; priorities (mwis_gdpg_call.py:211-216: float32 x float64 -> float64)
__global__ __launch_bounds__(kResBlock) void k_res_scatter(ResArgs a) {
    const int g = blockIdx.x;
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1];
    const bool act = a.active[g] != 0;
    for (int v = n0 + threadIdx.x; v < n1; v += kResBlock) {
        const bool alive = act && a.state[v] == 0;
        float s = 0.f;
        if (alive) s = a.scores_given ? a.scores[v] : a.sc[a.cidx[v]];
        if (a.scores && !a.scores_given) a.scores[v] = s;  // what a removed vertex (or a graph left alone) reports: 0
        double p = (double)s;
        if (a.predict_mwis && a.weights) p *= a.weights[v];
        a.prio[v] = alive ? p : 0.0;
    }
}

// ---- greedy_mode 1: the best-priority undecided vertex joins (np.argmax: lowest index among equals), its neighbours leave
__global__ __launch_bounds__(kResBlock) void k_res_central(ResArgs a) {
    __shared__ double sp[kResBlock];
    __shared__ int sv[kResBlock];
    __shared__ int s_bad;
    const int g = blockIdx.x;
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1], ng = n1 - n0;
    if (!a.active[g]) {
        if (threadIdx.x == 0) { if (a.rounds) a.rounds[g] = 0; if (a.totals) a.totals[g] = 0.0; }
        return;
    }
    if (threadIdx.x == 0) s_bad = 0;
    __syncthreads();
    double bp = 0.0;
    int bv = -1, bad = 0;
    for (int v = threadIdx.x; v < ng; v += kResBlock) {
        if (a.state[n0 + v] != 0) continue;
        const double p = a.prio[n0 + v];
        bad |= p != p;
        if (bv < 0 || p > bp) { bp = p; bv = v; }  // ascending v per thread: the first maximum stays
    }
    if (bad) s_bad = 1;
    sp[threadIdx.x] = bp;
    sv[threadIdx.x] = bv;
    __syncthreads();
    if (s_bad) {  // the reference's argmax would pick the NaN; like every other entry point: report, leave the graph alone
        if (threadIdx.x == 0) {
            atomicOr(a.status, DGCN_FAULT_NAN_PRIORITY);
            if (a.rounds) a.rounds[g] = -1;
            if (a.totals) a.totals[g] = 0.0;
        }
        return;
    }
    for (int off = kResBlock / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const double op = sp[threadIdx.x + off], mp = sp[threadIdx.x];
            const int ov = sv[threadIdx.x + off], mv = sv[threadIdx.x];
            if (ov >= 0 && (mv < 0 || op > mp || (op == mp && ov < mv))) { sp[threadIdx.x] = op; sv[threadIdx.x] = ov; }
        }
        __syncthreads();
    }
    const int c = sv[0];
    if (c < 0) return;  // (cannot happen for an active graph)
    const int rs = a.row_ptr[n0 + c], re = a.row_ptr[n0 + c + 1];
    for (int j = rs + threadIdx.x; j < re; j += kResBlock) {
        const int u = a.col_idx[j];
        if (u >= n0 && u < n1 && u != n0 + c && a.state[u] == 0) a.state[u] = 2;
    }
    if (threadIdx.x == 0) {
        a.state[n0 + c] = 1;
        if (a.rounds) a.rounds[g] = 1;
        if (a.totals) a.totals[g] = a.weights ? a.weights[n0 + c] : sp[0];
    }
}

// ---- greedy_mode 2, first launch: the first `beam` undecided vertices under (priority desc, index asc) - the stable
// argsort of -gcn_wts (mwis_gdpg_call.py:624-626).  Every wave selects ITS first `beam` (a thread keeps its <= 10 vertices'
// priorities in registers; `beam` rounds of: best of my vertices not taken yet, wave-wide argmax, its owner marks it taken);
// the waves' lists - sorted, and together they contain the graph's first `beam` - are ranked against each other: a candidate's
// rank is the sum over the lists of the entries ahead of it, a binary search per list, all 1 024 threads on (candidate, list) pairs.
// (Round 4's form counted, for EVERY undecided vertex, the vertices ahead of it - N^2 / 1 024 float64 compares per thread: 56 us
// per launch at 900 vertices, the largest single item of a rollout step on the any-size path.  First version of this one: the
// vertices' state and priority loaded one dependent pair after the other (18 us), the wave-wide argmax by __shfl_xor - six
// ds_bpermute round trips of three words through an LDS crossbar sixteen waves queue on (22 us for 16 rounds) -, the ranking a
// linear scan by 256 threads (20 us): 59 us, nothing gained.  Now: all loads in flight at once, the argmax inside a row of
// sixteen lanes by DPP moves and across the four rows by v_readlane, the ranking as above.)
__global__ __launch_bounds__(kResBlock) void k_res_cand(ResArgs a, int lds_nodes) {
    (void)lds_nodes;
    static_assert(kCandPer * kResBlock >= 9600, "a thread holds its vertices' priorities in registers");
    __shared__ __attribute__((aligned(8))) unsigned char scratch[cand_scratch_bytes(kResBlock)];
    __shared__ int s_bad;
    const int g = blockIdx.x;
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1], ng = n1 - n0;
    int32_t* cid = a.cid + (size_t)g * kMaxBeam;
    if (threadIdx.x < kMaxBeam) cid[threadIdx.x] = -1;
    if (!a.active[g]) return;
    if (threadIdx.x == 0) s_bad = 0;
    const int beam = min(a.beam, kMaxBeam);
    const int per = (ng + kResBlock - 1) / kResBlock;  // (uniform) my vertices: threadIdx.x + i * kResBlock, i < per
    double pv[kCandPer];
    uint8_t sv[kCandPer];
#pragma unroll
    for (int i = 0; i < kCandPer; ++i) {  // every load in flight before the first one is looked at
        const int v = (int)threadIdx.x + i * kResBlock;
        const bool in = i < per && v < ng;
        sv[i] = in ? a.state[n0 + v] : (uint8_t)1;
        pv[i] = in ? a.prio[n0 + v] : 0.0;
    }
    unsigned have = 0u;  // bit i: my vertex i is undecided and not selected yet
    int bad = 0;
#pragma unroll
    for (int i = 0; i < kCandPer; ++i)
        if (sv[i] == 0) {
            bad |= pv[i] != pv[i];
            have |= 1u << i;
        }
    __syncthreads();
    if (bad) s_bad = 1;
    __syncthreads();
    if (s_bad) {
        if (threadIdx.x == 0) atomicOr(a.status, DGCN_FAULT_NAN_PRIORITY);
        return;  // no candidates: k_res_pick leaves the graph alone
    }
    cand_select<kResBlock>(pv, have, per, beam, cid, scratch);
}

// ---- (instance i of graph g = the residual graph minus the closed neighbourhood of candidate i, mwis_gdpg_call.py:629-643:
// k_lgs makes that mask in its own LDS from the running state and the candidate list - LgsArgs::cand - so no launch writes
// beam x num_nodes mask bytes out first)

// ---- last launch: the candidate with the largest weight + completion total joins (totals within 1e-12 relative count
// as tied, the first one wins: np.isclose(cand, cand.max(), rtol=1e-12, atol=0), as in fused.hip)
__global__ __launch_bounds__(64) void k_res_pick(ResArgs a) {
    __shared__ int s_c;
    const int g = blockIdx.x;
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1];
    const int32_t* cid = a.cid + (size_t)g * kMaxBeam;
    {
        // a lane per candidate (one wave): its vertex, weight + completion total; the largest by a shuffle tree (a maximum: the
        // order of the compares does not matter), then the FIRST candidate within 1e-12 relative of it
        // (first form: thread 0 alone, sixteen dependent pairs of global round trips - 11 us per launch)
        const int beam = min(a.beam, kMaxBeam);
        const int lane = threadIdx.x;
        const bool on = a.active[g] != 0;
        const int my = (on && lane < beam) ? cid[lane] : -1;
        double cand = 0.0;
        if (my >= 0) cand = a.weights[n0 + my] + a.inst_totals[(size_t)lane * a.num_graphs + g];
        // candidates fill cid from the front: the first -1 ends the list (a lane behind a gap does not count)
        const unsigned long long valid = __ballot(my >= 0);
        const int nc = valid == ~0ull ? 64 : __ffsll((long long)~valid) - 1;
        const bool counts = lane < nc;
        double mx = counts ? cand : -1.0 / 0.0;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) mx = fmax(mx, __shfl_xor(mx, off));
        // (cand == mx: equal infinities are close for np.isclose; nobody close - NaN totals -: the first candidate, as in
        // fused.hip / tail.hip / rollout_bits.h: a step that picks nobody would leave the graph active for ever)
        const unsigned long long tied = __ballot(counts && (cand == mx || fabs(cand - mx) <= 1e-12 * fabs(mx)));
        int c = -1;
        if (nc > 0) c = __shfl(my, tied ? __ffsll((long long)tied) - 1 : 0);
        if (threadIdx.x == 0) {
            s_c = c;
            if (a.rounds) a.rounds[g] = c < 0 ? 0 : 1;
            if (a.totals) a.totals[g] = c < 0 ? 0.0 : a.weights[n0 + c];
        }
    }
    __syncthreads();
    const int c = s_c;
    if (c < 0) return;
    const int rs = a.row_ptr[n0 + c], re = a.row_ptr[n0 + c + 1];
    for (int j = rs + threadIdx.x; j < re; j += 64) {
        const int u = a.col_idx[j];
        if (u >= n0 && u < n1 && u != n0 + c && a.state[u] == 0) a.state[u] = 2;
    }
    if (threadIdx.x == 0) a.state[n0 + c] = 1;
}

// ---------------------------------------------------------------------------------------------------------------------
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Bump {
    char* p;
    size_t left;
    bool ok = true;
    template <typename T> T* take(size_t count) {
        const size_t nb = al256(count * sizeof(T));
        if (!ok || nb > left) { ok = false; return nullptr; }
        T* r = reinterpret_cast<T*>(p);
        p += nb;
        left -= nb;
        return r;
    }
};

static int model_in_dim(const DgcnModel* m) { return m->layers_host[0].in_dim; }

// the forward workspace of the layer-by-layer path (dgcn_gcn_forward_workspace(b, m, 0))
static size_t layered_bytes(const DgcnBatch* b, const DgcnModel* m) { return dgcn_gcn_forward_workspace(b, m, 0); }

// ---- [I, L, L.L] models (max_degree = 2: gcn/utils.py:268-271, gcn/layers.py:199-208; the shipped ..cheb2.. checkpoints) ----
// The second support is formed explicitly on the device (supports2.hip: SciPy's csr_matmat in float64, bit for bit) - the
// count pass, a scan, the fill pass - into a slice of the workspace sized by what the HOST knows without a round trip: a row
// of L.L has at most as many entries as its graph has vertices, so num_nodes * max_nodes entries always do.  The forward pass
// is the layer-by-layer one (forward.hip: Z = H.[W0 | W1 | W2], out = (Z0 + L.Z1) + L2.Z2), then the greedy step as for
// every other model: one C call, nothing returns to the host.
static size_t poly_cap2(const DgcnBatch* b) {
    return (size_t)std::max(b->num_nodes, 1) * (size_t)std::max(b->max_nodes, 1);
}
static bool poly_fits(const DgcnBatch* b) { return poly_cap2(b) < (size_t)0x7fffffff; }  // (int32 row pointers)

// the compact batch's ADJACENCY out of its support (k_res_fill writes L, diagonal first): what supports2.hip reads.
// arow2[v] = lrow[v] - (diagonal entries in front of row v); rows past the last vertex are empty in both.
__global__ __launch_bounds__(256) void k_lap_to_adj(const int32_t* __restrict__ lrow, const int32_t* __restrict__ lcol,
                                                    const int32_t* __restrict__ gptr2, int B, int num_nodes,
                                                    int32_t* __restrict__ arow2, int32_t* __restrict__ acol2) {
    const int live = gptr2[B];
    for (int v = blockIdx.x * 256 + threadIdx.x; v <= num_nodes; v += gridDim.x * 256) {
        const int rs = lrow[v];
        arow2[v] = rs - min(v, live);
        if (v < live) {
            const int re = lrow[v + 1], as = rs - v;
            for (int j = rs + 1; j < re; ++j) acol2[as + (j - rs - 1)] = lcol[j];
        }
    }
}

// L.L of batch `ab` (adjacency in ab->row_ptr / col_idx) into (l2row, l2col, l2val); the descriptor for the forward pass
static int poly_second_support(const DgcnBatch* ab, const double* dinv_table, int32_t table_len, int32_t* l2row, int32_t* l2col,
                               float* l2val, int32_t* status, DgcnCsr* out, hipStream_t s) {
    if (hipMemsetAsync(l2row, 0, ((size_t)ab->num_nodes + 1) * sizeof(int32_t), s) != hipSuccess)  // (rows no graph owns count 0)
        return fail(DGCN_ERR_LAUNCH, "dgcn_solve_batch: memset of the second support's row counts failed");
    if (int rc = dgcn_supports2_count_batch(ab, dinv_table, table_len, l2row, status, s)) return rc;
    if (int rc = dgcn_supports2_fill_batch(ab, dinv_table, table_len, l2row, l2col, l2val, status, s)) return rc;
    // (nnz is a hint for the SpMM's LDS carve-up only - a denser tile reads its rows from global memory, same chains)
    const size_t hint = std::min(poly_cap2(ab), (size_t)8 * ((size_t)ab->num_nodes + (size_t)std::max(ab->num_edges, 0)));
    *out = DgcnCsr{ab->num_nodes, (int32_t)hint, 0, l2row, l2col, l2val};
    return DGCN_OK;
}

size_t general_workspace(const DgcnBatch* b, const DgcnModel* m) {
    const size_t n = (size_t)std::max(b->num_nodes, 1), e = (size_t)std::max(b->num_edges, 0), B = (size_t)std::max(b->num_graphs, 1);
    size_t need = 256;                                                   // alignment slack
    need += 3 * al256(B * 4) + al256(B * 8) + 2 * al256((B + 1) * 4);      // na, ne, active, wmax, gptr2, eoff
    need += 2 * al256(n * 4);                                            // adeg, cidx
    need += al256((n + 1) * 4) + al256((n + e) * 4) + al256((n + e) * 4);  // the support (also the plain solve's)
    need += al256(n * (size_t)model_in_dim(m) * 4);                      // features of the compact batch
    need += al256(n * 4) + al256(n * 4);                                 // compact scores, full scores when the caller wants none
    need += al256(n * 8);                                                // priorities
    need += al256(layered_bytes(b, m)) + al256(std::max(big_workspace(b, m), big2_workspace(b, m)));
    need += al256(B * kMaxBeam * 4) + al256((size_t)kMaxBeam * n) + al256((size_t)kMaxBeam * B * 4) + al256((size_t)kMaxBeam * B * 8);
    if (m->num_supports == 3)  // the second support, and the compact batch's adjacency it is formed from
        need += 2 * al256((n + 1) * 4) + al256(e * 4 + 4) + al256(poly_cap2(b) * 4) + al256(poly_cap2(b) * 4);
    return need;
}

// which shapes the general path takes: [I, L] models of any widths (the layer-by-layer kernels), one score per vertex,
// graphs whose greedy-search state (14 bytes per vertex) fits one workgroup's LDS
int general_takes(const DgcnBatch* b, const DgcnModel* m) {
    if (!b || !m || !m->layers_host || (m->num_supports != 2 && m->num_supports != 3) || m->num_layers < 1) return 0;
    if (m->num_supports == 3 && !poly_fits(b)) return 0;
    if (m->layers_host[m->num_layers - 1].out_dim != 1) return 0;
    for (int l = 0; l < m->num_layers; ++l) {
        const DgcnLayer& L = m->layers_host[l];
        if (L.in_dim <= 0 || L.out_dim <= 0 || L.out_dim > 256 || !L.weights) return 0;
        if (l && L.in_dim != m->layers_host[l - 1].out_dim) return 0;
    }
    return b->max_nodes <= 9600;
}

// A1-A10 for any graph size: supports -> layer-by-layer forward -> local greedy search, three families of launches
int general_solve(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, const float* X,
                  float x_const, const double* weights, int32_t predict_mwis, float* scores, uint8_t* state, int32_t* rounds,
                  double* totals, int32_t* status, void* workspace, size_t workspace_bytes, hipStream_t s) {
    if (!general_takes(b, m))
        return fail(DGCN_ERR_UNSUPPORTED, "dgcn_solve_batch: model / graph shape outside both the fused kernel and the layer-by-layer path");
    const size_t n = (size_t)b->num_nodes, e = (size_t)b->num_edges;
    Bump w{reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255),
           workspace_bytes >= 256 ? workspace_bytes - 256 : 0};
    if (!workspace) w.ok = false;
    if (wide1_takes(b, m, X, 0)) {  // one- and two-layer models: supports, score, priority and greedy search in ONE launch (wide.hip)
        float* sc1 = scores ? scores : w.take<float>(n);
        if (!w.ok) return fail(DGCN_ERR_WORKSPACE, "dgcn_solve_batch: workspace of %zu bytes needed (dgcn_solve_workspace), got %zu",
                               general_workspace(b, m), workspace_bytes);
        return wide1_run(b, m, dinv_table, table_len, X, x_const, 0, weights, predict_mwis, 0, 0, 0, 0, sc1, state, rounds, totals, nullptr,
                         status, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0, s);
    }
    int32_t* lrow = w.take<int32_t>(n + 1);
    int32_t* lcol = w.take<int32_t>(n + e);
    float* lval = w.take<float>(n + e);
    float* sc = scores ? scores : w.take<float>(n);
    const size_t fbytes = layered_bytes(b, m);
    char* fws = w.take<char>(fbytes);
    const bool big2 = big2_takes(b, m) != 0;  // (977 .. 1 920 vertices; with option "big2" = 1 every shape k_big takes too)
    const bool big = !big2 && big_takes(b, m) != 0;
    char* bws = big ? w.take<char>(big_workspace(b, m)) : big2 ? w.take<char>(big2_workspace(b, m)) : nullptr;
    if (!w.ok) return fail(DGCN_ERR_WORKSPACE, "dgcn_solve_batch: workspace of %zu bytes needed (dgcn_solve_workspace), got %zu",
                           general_workspace(b, m), workspace_bytes);
    // constant input features on k_big's / k_big2's shapes: the whole path - supports, every layer, priority, greedy search - in ONE launch
    if (big && big_solve_takes(b, m, X))
        return big_solve(b, m, dinv_table, table_len, x_const, weights, predict_mwis, sc, state, rounds, totals, status, bws, s);
    if (big2 && big2_solve_takes(b, m, X))
        return big2_solve(b, m, dinv_table, table_len, x_const, weights, predict_mwis, sc, state, rounds, totals, status, bws, s);
    int rc = dgcn_supports_batch(b, dinv_table, table_len, lrow, lcol, lval, status, s);
    if (rc) return rc;
    DgcnCsr L = {b->num_nodes, (int32_t)(n + e), b->max_graph_edges + b->max_nodes, lrow, lcol, lval};
    DgcnCsr L2 = {};
    const DgcnCsr* sup[2] = {&L, &L2};
    if (m->num_supports == 3) {  // [I, L, L.L]
        int32_t* l2row = w.take<int32_t>(n + 1);
        int32_t* l2col = w.take<int32_t>(poly_cap2(b));
        float* l2val = w.take<float>(poly_cap2(b));
        if (!w.ok) return fail(DGCN_ERR_WORKSPACE, "dgcn_solve_batch: workspace of %zu bytes needed (dgcn_solve_workspace), got %zu",
                               general_workspace(b, m), workspace_bytes);
        if ((rc = poly_second_support(b, dinv_table, table_len, l2row, l2col, l2val, status, &L2, s))) return rc;
    }
    if ((rc = big ? big_forward(b, &L, m, X, x_const, sc, fws, bws, status, s)
              : big2 ? big2_forward(b, &L, m, X, x_const, sc, fws, bws, status, s) : layered_forward(b, sup, m, X, x_const, sc, fws, s))) return rc;
    return lgs_launch_common(b, nullptr, 0, sc, (predict_mwis && weights) ? weights : nullptr, nullptr, 1, 0, state, rounds, nullptr,
                             nullptr, weights, totals, status, s, nullptr, nullptr);
}

int general_residual(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, const float* X,
                     float x_const, int32_t feature_mode, const double* weights, int32_t predict_mwis, int32_t greedy_mode,
                     int32_t max_rounds, int32_t beam, int32_t options, float* scores, uint8_t* state, int32_t* rounds,
                     double* totals, int32_t* progress, int32_t* status, void* workspace, size_t workspace_bytes, hipStream_t s,
                     unsigned long long* tail_word, unsigned long long tail_tag) {
    if (!general_takes(b, m))
        return fail(DGCN_ERR_UNSUPPORTED, "dgcn_solve_residual_batch: model / graph shape outside both the fused kernel and the layer-by-layer path");
    const size_t n = (size_t)b->num_nodes, e = (size_t)b->num_edges, B = (size_t)b->num_graphs;
    const int cin = model_in_dim(m);
    const bool given = (options & DGCN_RESIDUAL_SCORES_GIVEN) != 0;
    Bump w{reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255),
           workspace_bytes >= 256 ? workspace_bytes - 256 : 0};
    if (!workspace) w.ok = false;
    ResArgs a = {};
    a.graph_ptr = b->graph_ptr; a.row_ptr = b->row_ptr; a.col_idx = b->col_idx;
    a.state = state; a.weights = weights; a.dinv = dinv_table; a.table_len = table_len;
    a.feature_mode = feature_mode; a.cin = cin; a.X = X;
    a.predict_mwis = predict_mwis; a.scores_given = given ? 1 : 0; a.beam = beam;
    a.by_priority = (options & DGCN_RESIDUAL_COMPLETE_BY_PRIORITY) ? 1 : 0;
    a.num_graphs = b->num_graphs; a.num_nodes = b->num_nodes;
    a.na = w.take<int32_t>(B); a.ne = w.take<int32_t>(B); a.active = w.take<int32_t>(B); a.wmax = w.take<double>(B);
    a.gptr2 = w.take<int32_t>(B + 1); a.eoff = w.take<int32_t>(B + 1);
    a.adeg = w.take<int32_t>(n); a.cidx = w.take<int32_t>(n);
    a.lrow = w.take<int32_t>(n + 1); a.lcol = w.take<int32_t>(n + e); a.lval = w.take<float>(n + e);
    a.Xc = (feature_mode == 1 || X) ? w.take<float>(n * (size_t)cin) : nullptr;
    float* sc = w.take<float>(n);
    a.sc = sc;
    a.scores = scores;
    a.prio = w.take<double>(n);
    const size_t fbytes = layered_bytes(b, m);
    char* fws = w.take<char>(fbytes);
    const bool big2 = big2_takes(b, m) != 0;
    const bool big = !big2 && big_takes(b, m) != 0;
    char* bws = big ? w.take<char>(big_workspace(b, m)) : big2 ? w.take<char>(big2_workspace(b, m)) : nullptr;
    a.cid = w.take<int32_t>(B * kMaxBeam);
    int32_t *arow2 = nullptr, *acol2 = nullptr, *l2row = nullptr, *l2col = nullptr;
    float* l2val = nullptr;
    if (m->num_supports == 3) {
        arow2 = w.take<int32_t>(n + 1);
        acol2 = w.take<int32_t>(e + 1);
        l2row = w.take<int32_t>(n + 1);
        l2col = w.take<int32_t>(poly_cap2(b));
        l2val = w.take<float>(poly_cap2(b));
    }
    uint8_t* inst_state = nullptr;
    int32_t* inst_rounds = nullptr;
    double* inst_totals = nullptr;
    if (greedy_mode == 2) {
        inst_state = w.take<uint8_t>((size_t)beam * n);
        inst_rounds = w.take<int32_t>((size_t)beam * B);
        inst_totals = w.take<double>((size_t)beam * B);
        a.inst_totals = inst_totals;
    }
    a.rounds = rounds; a.totals = totals; a.status = status; a.progress = progress;
    a.tail_word = tail_word; a.tail_tag = tail_tag;
    if (!w.ok) return fail(DGCN_ERR_WORKSPACE, "dgcn_solve_residual_batch: workspace of %zu bytes needed (dgcn_solve_workspace), got %zu",
                           general_workspace(b, m), workspace_bytes);
    const dim3 gb((unsigned)b->num_graphs), tb(kResBlock);
    bool wide = wide1_takes(b, m, X, feature_mode) != 0;
    bool cand_done = false;  // the rollout's candidates were selected inside the step's own launch (cand_select.h)
    int32_t whole_step = 0;  // ... and so were the completions and the pick (rollout_bits.h)
    if (!wide && big && big_residual_takes(b, m, X, feature_mode, options)) {
        // deep c32 stacks on graphs k_big takes, constant input features: activity test, the residual graph's support, every layer,
        // priorities and the greedy step (rounds / central pick) in ONE launch on the graph as it lies - no compaction, no k_lgs;
        // the rollout's remaining launches follow on the priorities it leaves (none when the step ran whole in that launch)
        const int rc = big_residual(b, m, dinv_table, table_len, x_const, weights, predict_mwis, greedy_mode, max_rounds, scores ? scores : sc,
                                    state, rounds, totals, progress, status, a.prio, a.active, a.cid, beam, a.by_priority, &whole_step,
                                    tail_word, tail_tag, bws, s);
        if (rc || greedy_mode != 2 || whole_step) return rc;  // (whole_step: completions and pick ran in that launch, rollout_bits.h)
        wide = true;  // (what follows is the same as behind the one-layer kernel: the rollout's launches)
        cand_done = true;  // (the candidates were selected at the end of that launch)
    } else if (!wide && big2 && big2_residual_takes(b, m, X, feature_mode, options)) {  // the same for 977 .. 1 920 vertices (k_big2)
        const int rc = big2_residual(b, m, dinv_table, table_len, x_const, weights, predict_mwis, greedy_mode, max_rounds, scores ? scores : sc,
                                     state, rounds, totals, progress, status, a.prio, a.active, a.cid, beam, a.by_priority, &whole_step,
                                     tail_word, tail_tag, bws, s);
        if (rc || greedy_mode != 2 || whole_step) return rc;
        wide = true;
        cand_done = true;
    } else
    if (wide) {
        // one-layer models: activity test, residual degrees, scores, priorities and the greedy step (rounds / central pick) in ONE
        // launch on the graph as it lies - no compaction; a rollout's completions and pick too (rollout_bits.h) unless k_lgs / k_res_pick have to follow
        const int rc = wide1_run(b, m, dinv_table, table_len, X, x_const, feature_mode, weights, predict_mwis, 1, given ? 1 : 0, greedy_mode,
                                 max_rounds, scores ? scores : sc, state, rounds, totals, progress, status, a.prio, a.active, a.cid, beam, a.by_priority,
                                 &whole_step, tail_word, tail_tag, s);
        if (rc || greedy_mode != 2 || whole_step) return rc;  // (whole_step: completions and pick ran in that launch, rollout_bits.h)
        cand_done = true;
    }
    if (!wide) {
        TimedLaunch t("general_prepare", s);
        DGCN_LAUNCH(t, k_res_count, gb, tb, 0, s, a);
        if (int rc = check_launch("k_res_count")) return rc;
    }
    if (!wide && !given) {
        {
            TimedLaunch t("general_prepare", s);
            DGCN_LAUNCH(t, k_res_scan, dim3(1), dim3(1024), 0, s, a);
            if (int rc = check_launch("k_res_scan")) return rc;
        }
        {
            TimedLaunch t("general_prepare", s);
            DGCN_LAUNCH(t, k_res_fill, gb, tb, 0, s, a);
            if (int rc = check_launch("k_res_fill")) return rc;
        }
        // the compact batch: same number of graphs, sizes bounded by the full batch's (the kernels read the real ones
        // from graph_ptr / row_ptr on the device; a graph that is left alone has no vertices there)
        DgcnBatch cb = *b;
        cb.graph_ptr = a.gptr2;
        cb.row_ptr = nullptr;
        cb.col_idx = nullptr;
        DgcnCsr L = {b->num_nodes, (int32_t)(n + e), b->max_graph_edges + b->max_nodes, a.lrow, a.lcol, a.lval};
        DgcnCsr L2 = {};
        const DgcnCsr* sup[2] = {&L, &L2};
        if (m->num_supports == 3) {
            // [I, L, L.L] on the residual graph: its adjacency out of the support k_res_fill wrote, then L.L as for a full batch
            {
                TimedLaunch t("general_prepare", s);
                const int blocks = (int)std::min<size_t>((n + 256) / 256, 1024);
                DGCN_LAUNCH(t, k_lap_to_adj, dim3(blocks), dim3(256), 0, s, a.lrow, a.lcol, a.gptr2, b->num_graphs, b->num_nodes, arow2, acol2);
                if (int rc = check_launch("k_lap_to_adj")) return rc;
            }
            DgcnBatch ab = cb;
            ab.row_ptr = arow2;
            ab.col_idx = acol2;
            if (int rc = poly_second_support(&ab, dinv_table, table_len, l2row, l2col, l2val, status, &L2, s)) return rc;
        }
        if (int rc = big ? big_forward(&cb, &L, m, a.Xc, x_const, sc, fws, bws, status, s)
                     : big2 ? big2_forward(&cb, &L, m, a.Xc, x_const, sc, fws, bws, status, s) : layered_forward(&cb, sup, m, a.Xc, x_const, sc, fws, s))
            return rc;
    }
    if (!wide) {
        TimedLaunch t("general_prepare", s);
        DGCN_LAUNCH(t, k_res_scatter, gb, tb, 0, s, a);
        if (int rc = check_launch("k_res_scatter")) return rc;
    }
    if (greedy_mode == 0)
        return lgs_launch_common(b, a.prio, 0, nullptr, nullptr, state, 1, max_rounds, state, rounds, nullptr, nullptr,
                                 weights /* null: the priorities */, totals, status, s, a.active, nullptr);
    if (greedy_mode == 1) {
        TimedLaunch t("general_greedy", s);
        DGCN_LAUNCH(t, k_res_central, gb, tb, 0, s, a);
        return check_launch("k_res_central");
    }
    if (!cand_done) {
        TimedLaunch t("general_greedy", s);
        DGCN_LAUNCH(t, k_res_cand, gb, tb, 0, s, a, 0);
        if (int rc = check_launch("k_res_cand")) return rc;
    }
    if (int rc = lgs_launch_common(b, a.by_priority ? a.prio : weights, 0, nullptr, nullptr, state, beam, 0, inst_state, inst_rounds,
                                   nullptr, nullptr, weights, inst_totals, status, s, a.active, a.cid))
        return rc;
    TimedLaunch t("general_greedy", s);
    DGCN_LAUNCH(t, k_res_pick, gb, dim3(64), 0, s, a);
    return check_launch("k_res_pick");
}

// -1 = automatic (the fused kernels where a graph's image fits, this path otherwise), 1 = always this path (tests, A/B runs)
// (option "general"; dgcn_set_general is the older name of the same word)
int general_setting() { return opt(OPT_GENERAL) > 0 ? 1 : -1; }

}  // namespace dgcn

extern "C" void dgcn_set_general(int32_t setting) { dgcn::opt_store(dgcn::OPT_GENERAL, setting > 0 ? 1 : -1); }
extern "C" int32_t dgcn_get_general(void) { return dgcn::general_setting(); }
