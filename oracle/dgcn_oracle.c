/*
 * dgcn_oracle.c - CPU twin of the HIP kernels.  TEST INFRASTRUCTURE ONLY (never linked into or
 * called by the product; used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).
 *
 * Same algorithm as oracle/ref_numpy.py (which restates the reference), but with the exact float32
 * operation ORDER the HIP kernels use, so that scores - and therefore selected sets - can be
 * compared bit for bit at full batch size:
 *   supports   (float)(-(dinv[deg u] * dinv[deg v])), diagonal 1.0f first in each row
 *              [gcn/utils.py:120-127, 258-274]
 *   transform  z[r][n] = fmaf chain over k = 0..cin-1 from 0           [gcn/layers.py:202]
 *              layer index 1 (the first product whose input is a hidden activation): the same chain carried
 *              in double (fma), rounded to float32 once
 *   aggregate  acc = G interleaved fmaf chains over the row's CSR entries + butterfly (orc_spmm);
 *              out = z0 + acc; out += bias;
 *              activation                                              [gcn/layers.py:206-216]
 *              layer index 0: one fma chain in double over the entries, out = (float)((double)z0 + acc [+ bias])
 *   (why those two: the first layer's output is an affine function of ONE scalar per vertex - its normalised
 *   degree sum, large on hubs - and the second layer's products cancel it; float32 rounding there is what the
 *   other 18 layers amplify.  profiles/r03_error_budget.txt)
 *   priority   (double)score * weight                                  [mwis_dqn_call.py:232]
 *   lgs        synchronous rounds, order (priority desc, index asc)    [heuristics.py:77-116]
 * Parity status: pinned for supports / lgs (golden vectors from the imported reference);
 * the GCN forward is "parity unpinned" at the TensorFlow boundary (see ref_numpy.py).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC -o _build/liboracle.so dgcn_oracle.c -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float act_apply(float x, int act) {
    if (act == 1) return x > 0.0f ? x : 0.2f * x;
    if (act == 2) return x > 0.0f ? x : 0.0f;
    return x;
}

/* L = I - D^-1/2 A D^-1/2, diagonal first.  Returns a DGCN_FAULT_* style bitmask (0 = ok). */
int orc_supports(int num_nodes, const int32_t* graph_ptr, int num_graphs, const int32_t* row_ptr,
                 const int32_t* col_idx, const double* dinv_table, int table_len, int32_t* lap_row_ptr,
                 int32_t* lap_col, float* lap_val) {
    int fault = 0;
    for (int g = 0; g < num_graphs; ++g) {
        const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1];
        for (int v = n0; v < n1; ++v) {
            const int rs = row_ptr[v], re = row_ptr[v + 1];
            const int deg = re - rs;
            double dv = 0.0;
            if (deg < table_len) dv = dinv_table[deg]; else fault |= 4;
            const int out = rs + v;
            lap_row_ptr[v] = out;
            lap_col[out] = v;
            lap_val[out] = 1.0f;
            for (int j = rs; j < re; ++j) {
                const int u = col_idx[j];
                float val = 0.0f;
                if (u < n0 || u >= n1) fault |= 8;
                else {
                    if (u == v) fault |= 1;
                    const int du = row_ptr[u + 1] - row_ptr[u];
                    double d = 0.0;
                    if (du < table_len) d = dinv_table[du]; else fault |= 4;
                    val = (float)(-(d * dv));
                }
                lap_col[j + v + 1] = u;
                lap_val[j + v + 1] = val;
            }
        }
    }
    lap_row_ptr[num_nodes] = row_ptr[num_nodes] + num_nodes;
    return fault;
}

/* T_2 = L.L as SciPy's csr_matmat forms it for gcn/utils.py:268-271 (float64 L with ascending columns):
 * row i: sums[k] += L[i,j] * L[j,k] for j ascending (multiply, then add; this file is built with
 * -ffp-contract=off), exact zeros dropped, ascending output columns, value cast to float32 (TF feed).
 * Pass col2 = NULL to count only.  row_ptr2[num_nodes+1] is written in both modes.  Adjacency rows must be
 * sorted.  Returns the fault mask. */
int orc_supports2(int num_nodes, const int32_t* graph_ptr, int num_graphs, const int32_t* row_ptr, const int32_t* col_idx,
                  const double* dinv_table, int table_len, int32_t* row_ptr2, int32_t* col2, float* val2) {
    int fault = 0, maxn = 0;
    for (int g = 0; g < num_graphs; ++g) if (graph_ptr[g + 1] - graph_ptr[g] > maxn) maxn = graph_ptr[g + 1] - graph_ptr[g];
    double* acc = (double*)malloc(((size_t)maxn + 1) * sizeof(double));
    double* dv = (double*)malloc(((size_t)maxn + 1) * sizeof(double));
    int out = 0;
    for (int g = 0; g < num_graphs; ++g) {
        const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1], ng = n1 - n0;
        for (int k = 0; k < ng; ++k) {
            const int deg = row_ptr[n0 + k + 1] - row_ptr[n0 + k];
            dv[k] = 0.0;
            if (deg < table_len) dv[k] = dinv_table[deg]; else fault |= 4;
        }
        for (int i = 0; i < ng; ++i) {
            for (int k = 0; k < ng; ++k) acc[k] = 0.0;
            const int rs = row_ptr[n0 + i], re = row_ptr[n0 + i + 1];
            int p = rs, diag_done = 0;
            while (p < re || !diag_done) {
                const int cj = p < re ? col_idx[p] - n0 : 0x7fffffff;
                int j;
                double lij;
                if (!diag_done && cj > i) { j = i; lij = 1.0; diag_done = 1; }
                else {
                    ++p;
                    if (cj < 0 || cj >= ng) { fault |= 8; continue; }
                    if (cj == i) { fault |= 1; continue; }
                    j = cj;
                    lij = -(dv[j] * dv[i]);
                }
                for (int q = row_ptr[n0 + j]; q < row_ptr[n0 + j + 1]; ++q) {
                    const int k = col_idx[q] - n0;
                    if (k < 0 || k >= ng || k == j) continue;
                    const double ljk = -(dv[j] * dv[k]);
                    acc[k] = acc[k] + lij * ljk;
                }
                acc[j] = acc[j] + lij * 1.0;
            }
            row_ptr2[n0 + i] = out;
            for (int k = 0; k < ng; ++k)
                if (acc[k] != 0.0) {
                    if (col2) { col2[out] = n0 + k; val2[out] = (float)acc[k]; }
                    ++out;
                }
        }
    }
    row_ptr2[num_nodes] = out;
    free(acc);
    free(dv);
    return fault;
}

void orc_transform(const float* H, int ldh, float h_const, int rows, int cin, const float* W, int ctot, float* Z,
                   int ldz) {
    for (int r = 0; r < rows; ++r)
        for (int n = 0; n < ctot; ++n) {
            float acc = 0.0f;
            for (int k = 0; k < cin; ++k) acc = fmaf(H ? H[(size_t)r * ldh + k] : h_const, W[k * ctot + n], acc);
            Z[(size_t)r * ldz + n] = acc;
        }
}

/* the same product with the chain carried in double (DGCN_PRECISE: layer index 1) */
void orc_transform_f64(const float* H, int ldh, float h_const, int rows, int cin, const float* W, int ctot, float* Z,
                       int ldz) {
    for (int r = 0; r < rows; ++r)
        for (int n = 0; n < ctot; ++n) {
            double acc = 0.0;
            for (int k = 0; k < cin; ++k) acc = fma((double)(H ? H[(size_t)r * ldh + k] : h_const), (double)W[k * ctot + n], acc);
            Z[(size_t)r * ldz + n] = (float)acc;
        }
}

/* Split factor of the row sum for feature width C: part of the arithmetic contract of
 * dgcn_spmm_batch (include/dgcn.h).  The shipped default is 1 for every width (plain sequential
 * chain); other factors exist only behind the DGCN_SPMM_SPLIT tuning knob. */
int orc_spmm_split(int C) {
    (void)C;
    return 1;
}

/* Row sum = G interleaved fmaf chains (entry i of the row -> chain i % G) + butterfly
 * "for off = G/2..1: p[g] += p[g ^ off]" (both partners compute the same commutative sum). */
void orc_spmm(int num_rows, const int32_t* row_ptr, const int32_t* col_idx, const float* values, const float* Z,
              int ldz, int C, const float* Y0, int ldy0, const float* bias, int act, float* Y, int ldy, int G) {
    if (G <= 0) G = orc_spmm_split(C);
    for (int v = 0; v < num_rows; ++v)
        for (int c = 0; c < C; ++c) {
            float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const int rs = row_ptr[v], re = row_ptr[v + 1];
            for (int j = rs; j < re; ++j) {
                const int g = (j - rs) % G;
                p[g] = fmaf(values[j], Z[(size_t)col_idx[j] * ldz + c], p[g]);
            }
            for (int off = G / 2; off >= 1; off >>= 1) {
                float q[8];
                for (int g = 0; g < G; ++g) q[g] = p[g] + p[g ^ off];
                for (int g = 0; g < G; ++g) p[g] = q[g];
            }
            float acc = p[0];
            if (Y0) acc = Y0[(size_t)v * ldy0 + c] + acc;
            if (bias) acc = acc + bias[c];
            Y[(size_t)v * ldy + c] = act_apply(acc, act);
        }
}

/* the same aggregation with ONE fma chain in double over the row's entries (DGCN_PRECISE: layer index 0):
 * out = (float)((double)y0 + acc [+ (double)bias]), then the activation in float32 */
void orc_spmm_f64(int num_rows, const int32_t* row_ptr, const int32_t* col_idx, const float* values, const float* Z,
                  int ldz, int C, const float* Y0, int ldy0, const float* bias, int act, float* Y, int ldy) {
    for (int v = 0; v < num_rows; ++v)
        for (int c = 0; c < C; ++c) {
            double acc = 0.0;
            for (int j = row_ptr[v]; j < row_ptr[v + 1]; ++j)
                acc = fma((double)values[j], (double)Z[(size_t)col_idx[j] * ldz + c], acc);
            if (Y0) acc = (double)Y0[(size_t)v * ldy0 + c] + acc;
            if (bias) acc = acc + (double)bias[c];
            Y[(size_t)v * ldy + c] = act_apply((float)acc, act);
        }
}

/* Whole forward, layer by layer, over the supports [I, T_1, .., T_k] (num_supports = k + 1; sup_* arrays hold
 * T_1..T_k).  dims[l], dims[l+1] = in/out of layer l; weights[l] is [in][num_supports*out] (W0 | W1 | ..);
 * biases[l] may be NULL; acts[l] activation code.  out = ((Z_0 + T_1.Z_1) + T_2.Z_2 ..) + b, activation. */
int orc_forward_poly(int num_nodes, int num_supports, const int32_t* const* sup_row_ptr, const int32_t* const* sup_col,
                     const float* const* sup_val, int num_layers, const int32_t* dims, const float* const* weights,
                     const float* const* biases, const int32_t* acts, const float* X, float x_const, float* scores) {
    int maxd = 0;
    const int K = num_supports;
    for (int l = 0; l <= num_layers; ++l) if (dims[l] > maxd) maxd = dims[l];
    float* Z = (float*)malloc((size_t)num_nodes * K * maxd * sizeof(float) + 16);
    float* Hb = (float*)malloc((size_t)num_nodes * maxd * sizeof(float) + 16);
    float* Tb = (float*)malloc((size_t)num_nodes * maxd * sizeof(float) + 16);
    if (!Z || !Hb || !Tb) { free(Z); free(Hb); free(Tb); return -1; }
    const float* H = X;
    int ldh = dims[0];
    for (int l = 0; l < num_layers; ++l) {
        const int cin = dims[l], cout = dims[l + 1], ctot = K * cout;
        if (l == 1) orc_transform_f64(H, ldh, x_const, num_nodes, cin, weights[l], ctot, Z, ctot);
        else orc_transform(H, ldh, x_const, num_nodes, cin, weights[l], ctot, Z, ctot);
        float* out = (l == num_layers - 1) ? scores : Hb;
        const float* run = Z;
        int ldrun = ctot;
        for (int i = 1; i < K; ++i) {
            const int fin = i == K - 1;
            float* dst = fin ? out : Tb;
            if (l == 0)
                orc_spmm_f64(num_nodes, sup_row_ptr[i - 1], sup_col[i - 1], sup_val[i - 1], Z + i * cout, ctot, cout, run,
                             ldrun, fin ? biases[l] : NULL, fin ? acts[l] : 0, dst, cout);
            else
                orc_spmm(num_nodes, sup_row_ptr[i - 1], sup_col[i - 1], sup_val[i - 1], Z + i * cout, ctot, cout, run, ldrun,
                         fin ? biases[l] : NULL, fin ? acts[l] : 0, dst, cout, 0);
            run = dst;
            ldrun = cout;
        }
        H = out;
        ldh = cout;
    }
    free(Z);
    free(Hb);
    free(Tb);
    return 0;
}

int orc_forward(int num_nodes, const int32_t* lap_row_ptr, const int32_t* lap_col, const float* lap_val,
                int num_layers, const int32_t* dims, const float* const* weights, const float* const* biases,
                const int32_t* acts, const float* X, float x_const, float* scores) {
    const int32_t* rp[1] = {lap_row_ptr};
    const int32_t* cl[1] = {lap_col};
    const float* vl[1] = {lap_val};
    return orc_forward_poly(num_nodes, 2, rp, cl, vl, num_layers, dims, weights, biases, acts, X, x_const, scores);
}

void orc_priority(int n, const float* scores, const double* weights, double* prio) {
    for (int v = 0; v < n; ++v) prio[v] = weights ? (double)scores[v] * weights[v] : (double)scores[v];
}

/* Local greedy search over a batch; outputs as dgcn_lgs_batch.  Returns fault mask. */
int orc_lgs(int num_graphs, const int32_t* graph_ptr, const int32_t* row_ptr, const int32_t* col_idx,
            const double* prio, int max_rounds, uint8_t* state, int32_t* rounds, int64_t* stats, int32_t* overhead,
            const double* sum_weights, double* totals) {
    int fault = 0;
    int maxn = 0;
    for (int g = 0; g < num_graphs; ++g) if (graph_ptr[g + 1] - graph_ptr[g] > maxn) maxn = graph_ptr[g + 1] - graph_ptr[g];
    uint8_t* win = (uint8_t*)malloc((size_t)maxn + 1);
    for (int g = 0; g < num_graphs; ++g) {
        const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1];
        int bad = 0;
        for (int v = n0; v < n1; ++v) { state[v] = 0; if (overhead) overhead[v] = 0; if (prio[v] != prio[v]) bad = 1; }
        if (bad) {
            fault |= 2;
            if (rounds) rounds[g] = -1;
            if (stats) { stats[2 * g] = 0; stats[2 * g + 1] = 0; }
            if (totals) totals[g] = 0.0;
            continue;
        }
        int remaining = n1 - n0, r = 0;
        int64_t p2p = 0, bst = 0;
        while (remaining > 0 && (max_rounds <= 0 || r < max_rounds)) {
            bst += remaining;
            for (int v = n0; v < n1; ++v) {
                win[v - n0] = 0;
                if (state[v]) continue;
                int lost = 0, resid = 0;
                for (int j = row_ptr[v]; j < row_ptr[v + 1]; ++j) {
                    const int u = col_idx[j];
                    if (state[u] == 0) {
                        ++resid;
                        if (prio[u] > prio[v] || (prio[u] == prio[v] && u < v)) lost = 1;
                    }
                }
                p2p += resid;
                if (overhead) overhead[v] += resid + ((!lost && resid > 0) ? 1 : 0);
                win[v - n0] = !lost;
            }
            for (int v = n0; v < n1; ++v)
                if (win[v - n0])
                    for (int j = row_ptr[v]; j < row_ptr[v + 1]; ++j)
                        if (state[col_idx[j]] == 0) state[col_idx[j]] = 2;
            remaining = 0;
            for (int v = n0; v < n1; ++v) {
                if (win[v - n0]) state[v] = 1;
                else if (state[v] == 0) ++remaining;
            }
            ++r;
        }
        int members = 0;
        double tot = 0.0;
        for (int v = n0; v < n1; ++v)
            if (state[v] == 1) { ++members; tot += sum_weights ? sum_weights[v] : prio[v]; }
        if (rounds) rounds[g] = r;
        if (stats) { stats[2 * g] = p2p; stats[2 * g + 1] = bst + members; }
        if (totals) totals[g] = tot;
    }
    free(win);
    return fault;
}
