/*
 * dgcn.h - C ABI of libdgcn.so: the MI355X (gfx950) implementation of distgcn's
 * GCN-forward + local-greedy MWIS hot path.
 *
 * The reference (zhongyuanzhao/distgcn) is pure Python and has no native ABI; each entry point
 * below names the reference Python interface it replaces (file:line relative to the reference
 * root).  The Python shim in distgcn_amd/ binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller unless its name ends in _host.
 *    The library allocates nothing per call and keeps no caller pointer after returning.
 *  - All launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *    stream).  No entry point synchronises the device.
 *  - Return value: 0 = ok, < 0 = error; dgcn_last_error() gives the thread-local message.
 *  - A batch of graphs is ONE block-diagonal CSR: graph g owns global node ids
 *    [graph_ptr[g], graph_ptr[g+1]); col_idx holds GLOBAL node ids; indices are int32.
 *    The adjacency is symmetric, has no self-loops and implicit values 1.0 (what
 *    Data_Generation.py:218 stores and heuristics.py:77-116 requires).
 *  - Data-dependent faults (self-loop, NaN priority, degree beyond the table) cannot be
 *    returned synchronously; kernels OR a bit into the caller's device word `status`
 *    (DGCN_FAULT_*), which the caller reads together with the results.
 */
#ifndef DGCN_H
#define DGCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGCN_VERSION 100 /* 0.1.0 */

enum { DGCN_OK = 0, DGCN_ERR_ARG = -1, DGCN_ERR_LAUNCH = -2, DGCN_ERR_UNSUPPORTED = -3, DGCN_ERR_WORKSPACE = -4 };

/* activation codes (gcn/layers.py:216; tf.nn.leaky_relu alpha = 0.2, gcn/models.py:553) */
enum { DGCN_ACT_LINEAR = 0, DGCN_ACT_LEAKY_RELU = 1, DGCN_ACT_RELU = 2 };

/* bits OR-ed into *status by kernels */
enum {
    DGCN_FAULT_SELF_LOOP = 1,    /* adjacency has a diagonal entry (heuristics.py:94 would loop forever) */
    DGCN_FAULT_NAN_PRIORITY = 2, /* NaN priority (heuristics.py:103-111 never selects it: infinite loop) */
    DGCN_FAULT_DEGREE_RANGE = 4, /* vertex degree >= dinv_table length, or a row with (many) more entries than its graph has vertices */
    DGCN_FAULT_BAD_COLUMN = 8,   /* column id outside the owning graph's node range */
    DGCN_FAULT_CLUSTER = 16      /* small batches only (one graph on several workgroups): the workgroups of a graph did not
                                    end up on one XCD, or one of them never arrived - results are not valid; rerun after
                                    dgcn_set_option("fused_cluster", 0) */
};

/* Block-diagonal adjacency of a batch (host struct holding device pointers). */
typedef struct DgcnBatch {
    int32_t num_graphs;       /* B */
    int32_t num_nodes;        /* sum of N_g */
    int32_t num_edges;        /* sum of nnz(A_g) (directed count, = 2 x undirected edges) */
    int32_t max_nodes;        /* max N_g (sizes LDS tiles and launch geometry) */
    int32_t max_graph_edges;  /* max nnz(A_g) over the batch (sizes the LDS copy of one graph's columns) */
    const int32_t* graph_ptr; /* [B+1]   node offsets */
    const int32_t* row_ptr;   /* [num_nodes+1] */
    const int32_t* col_idx;   /* [num_edges]   global node ids */
} DgcnBatch;

/* A support matrix S (T1 = L = I - D^-1/2 A D^-1/2) in CSR, block-diagonal like the batch.
 * Row v stores its diagonal entry FIRST, then the neighbours in adjacency order:
 * row_ptr[v] = batch.row_ptr[v] + v, nnz = num_edges + num_nodes. */
typedef struct DgcnCsr {
    int32_t num_rows;
    int32_t nnz;
    int32_t max_graph_nnz;  /* largest per-graph nnz (sizes the LDS copy of one graph's entries); 0 = unknown */
    const int32_t* row_ptr; /* [num_rows+1] */
    const int32_t* col_idx; /* [nnz] global ids */
    const float* values;    /* [nnz] */
} DgcnCsr;

/* One GraphConvolution layer (gcn/layers.py:149-216). */
typedef struct DgcnLayer {
    int32_t in_dim;
    int32_t out_dim;
    const float* weights; /* [in_dim][num_supports * out_dim] row-major: column block i is weights_i of
                             layers.py:174-182, so one product H.weights yields every pre_sup of :202 */
    const float* bias;    /* [out_dim] or NULL (layers.py:183-184) */
    int32_t act;          /* DGCN_ACT_* */
} DgcnLayer;

/* GCN_DQN / GCN2_DQN layer stack (gcn/models.py:536-573, 670-708). */
typedef struct DgcnModel {
    int32_t num_layers;
    int32_t num_supports;          /* 1 + max_degree: 2 ([I, L]) or 3 ([I, L, L.L]; layer-by-layer path only) */
    const DgcnLayer* layers_host;  /* HOST array of num_layers descriptors */
} DgcnModel;

int dgcn_version(void);
const char* dgcn_last_error(void);

/* ---- Options: the library's only process-wide state ------------------------------------------------------
 * The library reads NO environment variable.  Every switch between code paths (all of which give the same results: they
 * exist for A/B measurements, for the tests' witnesses - "the same batch down the other path, bit for bit" - and for the
 * fault recovery of the small-batch launch) is one word of this table: set / get are atomic, take effect with the next call
 * of any thread, and cost a launch nothing.  -1 = automatic (the default of every path switch).  Keys (csrc/options.h has
 * the full list with defaults; dgcn_option_name enumerates it):
 *   "fused_cluster" (-1 | 0 | K), "fused_block" (512 | 1024), "fused_order", "fused_fold", "fused_gw", "narrow_pad", "general" (1: every shape down the any-size path), "shallow", "shallow_long", "shallow_block",
 *   "wide1", "wide2", "wide_ahead", "rollout_bits", "big", "big_solve", "big_block", "big_tiles", "big_residual", "big2",
 *   "tail", "layer_fuse", "lgs_lpv", "lgs_block", "spmm_*" (tuning of the stand-alone SpMM), "host_direct_bytes",
 *   "host_compact", "host_compact_direct", "host_done_word", "test_cluster_fault" (test hook: the small-batch launch reports
 *   DGCN_FAULT_CLUSTER although there is none), "diag_flags" / "diag_stamps" (read by -DDGCN_DIAG builds only).
 * Unknown key: DGCN_ERR_ARG.  The Python package applies DGCN_OPTIONS="key=value,key=value" from ITS environment once at
 * load (distgcn_amd/_lib.py) - a convenience of the host layer, not of this library. */
int dgcn_set_option(const char* key, int64_t value);
int dgcn_get_option(const char* key, int64_t* value);
int dgcn_option_count(void);
const char* dgcn_option_name(int32_t index, int64_t* default_value /* or NULL */); /* NULL past the end */

/* ---- batch ingestion (host side): what mwis_dqn_test.py:304-321 does one .mat file at a time ------------
 * Packs num_graphs per-graph CSR adjacencies (SciPy's indptr / indices arrays as they are, int32 or int64:
 * index_bytes = 4 / 8; optional float64 vertex weights) into ONE block-diagonal batch laid out in the
 * caller's HOST staging buffer (pinned memory, so that a single hipMemcpyAsync moves the batch):
 *   [graph_ptr int32[B+1] | row_ptr int32[N+1] | col_idx int32[E] (global ids) | weights float64[N]]
 * each section 16-byte aligned at the byte offsets returned in DgcnPackInfo; the same offsets hold on the
 * device copy.  dgcn_pack_measure reads only the sizes (fills everything but max_degree) so the caller can
 * size the buffer; dgcn_pack_batch writes it with num_threads workers (0 = up to 8) and fills max_degree.
 * Structural validation only (what could make a kernel read out of bounds: indptr monotone from 0, column
 * ids inside [0, n)); self-loops and NaNs are reported later by the kernels' status word.
 * All pointers here are HOST pointers.  Thread-safe; keeps no pointer after returning. */
typedef struct DgcnPackInfo {
    int32_t num_graphs, num_nodes, num_edges, max_nodes, max_graph_edges, max_degree;
    int64_t off_graph_ptr, off_row_ptr, off_col_idx, off_weights; /* byte offsets; off_weights = -1 without weights */
    int64_t total_bytes;
} DgcnPackInfo;
int dgcn_pack_measure(const void* const* indptr_host, const int32_t* num_nodes_host, int32_t num_graphs,
                      int32_t index_bytes, int32_t with_weights, DgcnPackInfo* info,
                      int64_t* nnz_out_host /* [num_graphs] entries per graph, or NULL: lets the caller check its indices arrays' lengths */);
int dgcn_pack_batch(const void* const* indptr_host, const void* const* indices_host, const double* const* weights_host,
                    const int32_t* num_nodes_host, int32_t num_graphs, int32_t index_bytes,
                    void* staging_host, size_t staging_bytes, DgcnPackInfo* info, int32_t num_threads);

/* The compact TRANSFER format of the same batch: 16-bit LOCAL column ids and a 16-bit degree per vertex instead of 32-bit
 * global ids and row pointers - what crosses PCIe (and what the packing loop has to write) is about half the bytes
 * (C3: 5.0 MB instead of 9.2):
 *   [graph_ptr int32[B+1] | edge_ptr int32[B+1] (entries before graph g) | deg uint16[N] | col uint16[E] | weights float64[N]]
 * dgcn_pack_compact_layout: offsets from the ordinary DgcnPackInfo (dgcn_pack_measure).  dgcn_pack_compact_batch: writes it
 * with the same validation as dgcn_pack_batch; returns 0, < 0 (structural error), or 1 = "not compactable" (a graph of more
 * than 65 535 vertices, a vertex of more than 65 535 neighbours): pack the ordinary format then; fills info->max_degree.
 * dgcn_expand_compact_batch (device, one launch): rebuilds row_ptr[num_nodes + 1] and col_idx[num_edges] - entry for entry
 * what dgcn_pack_batch writes for the same graphs, entry order included; graph_ptr and weights are used where they lie in
 * the compact buffer.  dgcn_host_solver_* does all of this by itself for batches it copies to the device. */
typedef struct DgcnCompactInfo {
    int64_t off_graph_ptr, off_edge_ptr, off_deg, off_col, off_weights; /* byte offsets; off_weights = -1 without weights */
    int64_t total_bytes;
} DgcnCompactInfo;
int dgcn_pack_compact_layout(const DgcnPackInfo* info, DgcnCompactInfo* compact);
int dgcn_pack_compact_batch(const void* const* indptr_host, const void* const* indices_host, const double* const* weights_host,
                            const int32_t* num_nodes_host, int32_t num_graphs, int32_t index_bytes,
                            void* staging_host, size_t staging_bytes, DgcnPackInfo* info, const DgcnCompactInfo* compact,
                            int32_t num_threads);
int dgcn_expand_compact_batch(const void* compact_dev, const DgcnCompactInfo* compact, int32_t num_graphs, int32_t num_nodes,
                              int32_t max_nodes, int32_t* row_ptr_out, int32_t* col_idx_out, void* stream);

/* ---- A1/A2: gcn/utils.py:120-127 normalize_adj + :258-274 simple_polynomials (k = 1) ----------
 * Builds L = I - D^-1/2 A D^-1/2 for the whole batch.  dinv_table[d] must hold the float64 value
 * numpy.power(d, -0.5) with inf -> 0 (table built once by the host so the float64 bits equal the
 * reference's); each off-diagonal is (float)(-(dinv[deg u] * dinv[deg v])), the diagonal 1.0f -
 * exactly the float32 the reference feeds TF after its float64 -> float32 cast.
 * Outputs (caller-allocated): lap_row_ptr[num_nodes+1], lap_col[num_edges+num_nodes],
 * lap_val[num_edges+num_nodes].  Faults: SELF_LOOP, DEGREE_RANGE, BAD_COLUMN. */
int dgcn_supports_batch(const DgcnBatch* batch, const double* dinv_table, int32_t table_len,
                        int32_t* lap_row_ptr, int32_t* lap_col, float* lap_val,
                        int32_t* status, void* stream);

/* ---- A2 with k = 2: gcn/utils.py:268-271 "t_new = t_k[-1]*laplacian" (max_degree = 2: the shipped
 * result_IS4SAT_deep_ld1_c1_l{1,2}_cheb2_* checkpoints) --------------------------------------------------
 * T_2 = L.L formed explicitly, as SciPy's csr_matmat forms it on the float64 L with sorted columns: for
 * output row i the products L[i,j]*L[j,k] are added in float64 in ascending order of j (multiply and add
 * rounded separately), sums that are exactly 0 are dropped, the result is cast to float32 (TF's feed) -
 * bit-identical to the reference's values (tests/golden/supports.npz *_lap2_*).  Rows come out with
 * ascending columns (global ids).  Two calls because the CALLER allocates:
 *   count: lap2_row_ptr[num_nodes+1] <- row starts (exclusive scan of the row lengths), [num_nodes] = nnz
 *   (read lap2_row_ptr[num_nodes] back, allocate lap2_col / lap2_val)
 *   fill:  writes the entries.
 * The adjacency rows must have ascending columns for the bit-exact order (otherwise the values differ
 * by float64 rounding only).  Graphs up to 9 600 vertices.  Faults as dgcn_supports_batch. */
int dgcn_supports2_count_batch(const DgcnBatch* batch, const double* dinv_table, int32_t table_len,
                               int32_t* lap2_row_ptr, int32_t* status, void* stream);
int dgcn_supports2_fill_batch(const DgcnBatch* batch, const double* dinv_table, int32_t table_len,
                              const int32_t* lap2_row_ptr, int32_t* lap2_col, float* lap2_val,
                              int32_t* status, void* stream);

/* ---- K4 (+K5-K7): gcn/layers.py:206 sparse_tensor_dense_matmul, :208 add_n, :211 bias, :216 act
 * Y[v, 0:C] = act( Y0[v, 0:C] + sum_j S.values[j] * Z[S.col_idx[j], 0:C] + bias[0:C] )
 * for every row v of the block-diagonal S.  Summation order (part of the contract, mirrored by
 * oracle/dgcn_oracle.c): with G = dgcn_spmm_split(C), entry number i of the row goes to partial sum
 * i % G; each partial sum is a float32 fmaf chain in CSR order starting from 0; the partials are
 * combined by the butterfly "for off = G/2..1: p[g] += p[g ^ off]"; then "Y0 + sum", then "+ bias".
 * G depends on C only.  Y0 and bias may be NULL (plain SpMM: K4 alone).
 * ldz / ldy / ldy0 are row strides in floats.  graph_ptr/num_graphs/max_nodes let a workgroup
 * keep its graph's slice of Z in LDS; pass graph_ptr = NULL to force the global-gather path. */
int dgcn_spmm_split(int32_t C);
int dgcn_spmm_batch(const DgcnCsr* S, const int32_t* graph_ptr, int32_t num_graphs, int32_t max_nodes,
                    const float* Z, int32_t ldz, int32_t C,
                    const float* Y0, int32_t ldy0, const float* bias, int32_t act,
                    float* Y, int32_t ldy, void* stream);

/* The same aggregation with the row sum carried in double: ONE fma chain over the row's entries in CSR order starting
 * from 0.0, then (double)Y0 + sum, then + (double)bias, rounded to float32 once, activation in float32.  This is the
 * contract of LAYER INDEX 0 in every forward / solve entry point below (see "Precision" at dgcn_gcn_forward_batch). */
int dgcn_spmm_f64acc_batch(const DgcnCsr* S, const int32_t* graph_ptr, int32_t num_graphs, int32_t max_nodes,
                           const float* Z, int32_t ldz, int32_t C,
                           const float* Y0, int32_t ldy0, const float* bias, int32_t act,
                           float* Y, int32_t ldy, void* stream);

/* ---- K2/K3: gcn/layers.py:29-31 dot(x, W_i) for all supports at once -------------------------
 * Z[r, 0:ctot] = sum_k H[r, k] * W[k, 0:ctot], a float32 fmaf chain over k = 0..cin-1 starting
 * from 0 (bit-identical between the MFMA and the VALU code paths).  W is [cin][ctot] row-major
 * (the per-support weights_i concatenated along columns).  H == NULL means every entry of H
 * equals h_const (the reference's row-normalised all-ones features, gcn/utils.py:98-106). */
int dgcn_transform_batch(const float* H, int32_t ldh, float h_const, int32_t rows, int32_t cin,
                         const float* W, int32_t ctot, float* Z, int32_t ldz, void* stream);

/* The same product with the k chain carried in double (fma from 0.0, ascending k) and rounded to float32 once: the
 * contract of LAYER INDEX 1 in every forward / solve entry point below. */
int dgcn_transform_f64acc_batch(const float* H, int32_t ldh, float h_const, int32_t rows, int32_t cin,
                                const float* W, int32_t ctot, float* Z, int32_t ldz, void* stream);

/* ---- A4-A6: sess.run(model.outputs_softmax) for a batch (mwis_dqn_call.py:140-143) ------------
 * Precision (part of the ABI, mirrored by oracle/dgcn_oracle.c, identical in every mode and entry point): float32
 * fmaf chains as described at dgcn_transform_batch / dgcn_spmm_batch, EXCEPT the aggregation of layer index 0 and
 * the transform of layer index 1, whose chains run in double and are rounded once (dgcn_spmm_f64acc_batch,
 * dgcn_transform_f64acc_batch).  Reason: the first layer's output is an affine function of one scalar per vertex (its
 * degree-normalised neighbour sum, large on hub vertices) and the second layer's products cancel it; float32 rounding
 * in those two places is what the remaining layers amplify.  Measured on 4 000 BA graphs with the shipped 20-layer
 * model: max |score - float64 evaluation| 1.8e-5 with float32 everywhere, 5.5e-6 with these two chains in double
 * (a NumPy float32 evaluation of the reference's formula: 1.4e-5).  DESIGN.md section 3.
 * What this means for a caller who compares with float32 scores from elsewhere (TensorFlow's, NumPy's): on the BASELINE
 * configurations the scores are within 1e-5 (absolute) of the float64 evaluation on every graph; they are within 1e-5 of
 * the NumPy float32 evaluation too EXCEPT on 9 of the 4 000 BA graphs at l = 20 (up to 1.61e-5: hub-heavy N = 300, m = 2
 * graphs, ids in tests/test_full_size_parity.py) - where seven float32 summation orders of the same formula differ from
 * one another by up to 2.3e-5 (profiles/r04_f32_order_envelope.json): no float32 result is "the" reference there.  The
 * selected sets are identical under every order of that envelope, on all 9 064 graph evaluations.
 *
 * scores[num_nodes * out_dim] = GCN forward over the batch.  X is the dense row-normalised
 * feature matrix [num_nodes][in_dim] or NULL for "every entry = x_const".
 * workspace: dgcn_gcn_forward_workspace() bytes of device scratch.
 * mode: 0 = layer-by-layer (transform + SpMM kernels), 1 = fused per-graph persistent kernel. */
size_t dgcn_gcn_forward_workspace(const DgcnBatch* batch, const DgcnModel* model, int32_t mode);
int dgcn_gcn_forward_batch(const DgcnBatch* batch, const DgcnCsr* lap, const DgcnModel* model,
                           const float* X, float x_const, float* scores,
                           void* workspace, size_t workspace_bytes, int32_t mode, void* stream);

/* The same forward for models with more than two supports (gcn/layers.py:199-208 sums over ALL supports):
 * supports_host is a HOST array of model->num_supports - 1 CSR matrices T_1 = L, T_2 = L.L (T_0 = I is
 * implicit).  Per layer Z = H.[W_0 | W_1 | W_2], out = act(((Z_0 + T_1.Z_1) + T_2.Z_2) + b): tf.add_n adds
 * left to right.  Layer by layer only (workspace: dgcn_gcn_forward_workspace(batch, model, 0)). */
int dgcn_gcn_forward_poly_batch(const DgcnBatch* batch, const DgcnCsr* const* supports_host, const DgcnModel* model,
                                const float* X, float x_const, float* scores,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ---- output heads of the model classes ---------------------------------------------------------------------
 * dual: GCN2_DQN(is_dual=True), gcn/models.py:651-653 - per graph,
 *       out[v][j-1] = mean_v(act[v][0]) + (act[v][j] - mean_v(act[v][j])), j = 1..out_dim-1; out is [num_nodes][out_dim-1].
 * skip: GCN_DQN with FLAGS.skip, gcn/models.py:505-521 - out = concat([X, act], axis 1) . kernel + bias
 *       (tf.layers.dense: kernel [in_dim + out_dim][out_dim], bias [out_dim] or NULL); X NULL = constant features. */
int dgcn_head_dual_batch(const float* act, int32_t out_dim, const int32_t* graph_ptr, int32_t num_graphs,
                         float* out, void* stream);
int dgcn_head_skip_batch(const float* X, float x_const, int32_t in_dim, const float* act, int32_t out_dim,
                         const float* kernel, const float* bias, int32_t rows, float* out, void* stream);

/* ---- models.py:526/660  pred = argmax(outputs, axis 0), per graph, first maximum wins ---------*/
int dgcn_argmax_batch(const float* scores, int32_t ld, const int32_t* graph_ptr, int32_t num_graphs,
                      int32_t* arg_out, void* stream);

/* ---- A7 + A8/A8'/A9: heuristics.py:77-305 local_greedy_search{,_count,_stats,_overhead,_nstep}
 * and heuristics.py:13-35 greedy_search (same set under the (w desc, index asc) order) ----------
 * Priority of vertex v: prio[v] if prio != NULL, else (double)scores[v] * weights[v]
 * (mwis_dqn_call.py:232, f32 x f64 -> f64) or (double)scores[v] when weights == NULL.
 * Rounds run until no vertex remains or max_rounds (> 0) rounds have run (_nstep, :280).
 * Outputs (any may be NULL except state):
 *   state[num_nodes]    1 = in the set, 2 = excluded as neighbour of a member (nb_is), 0 = still remaining
 *   rounds[B]           rounds executed (:160)
 *   stats[B][2]         {p2p, bst} of _stats (:184-208)
 *   overhead[num_nodes] per-vertex message count of _overhead (:236-262)
 *   totals[B]           sum of sum_weights[v] over members (:115; float64)
 * Faults: NAN_PRIORITY (that graph is left untouched: state 0, rounds -1). */
int dgcn_lgs_batch(const DgcnBatch* batch, const double* prio, const float* scores, const double* weights,
                   int32_t max_rounds, uint8_t* state, int32_t* rounds, int64_t* stats, int32_t* overhead,
                   const double* sum_weights, double* totals, int32_t* status, void* stream);

/* ---- how fragile is a selected set against score error? ---------------------------------
 * (mwis_gdpg_call.py:211-216 feeds float32 scores x float64 weights into heuristics.py:103-111's compares.)
 * The search returns the lexicographically-first maximal independent set S under (priority desc, index asc),
 * which is the unique independent set in which every excluded vertex has a member neighbour ahead of it.
 * risky[g] = number of excluded vertices v (state 2) of graph g WITHOUT a member neighbour u (state 1) with
 *     p_u - p_v > delta * (|w_u| + |w_v|)        (p = prio, or (double)score * weight; |w| = 1 when prio is given
 *                                                 or weights is NULL)
 * risky[g] == 0 proves graph g's set is identical for EVERY score vector within delta (per score) of this one -
 * e.g. TensorFlow's float32 result if it is within delta of these scores.  `state` must be a finished search. */
int dgcn_margin_risk_batch(const DgcnBatch* batch, const double* prio, const float* scores, const double* weights,
                           const uint8_t* state, double delta, int32_t* risky, void* stream);

/* ---- greedy search on many residuals of the same batch: the rollout of mwis_gdpg_call.py:629-645 ----
 * Instance k (0 <= k < num_instances) runs the local greedy search on the batch with the vertices
 * whose init_state[k][v] != 0 taken out beforehand - exactly greedy_search(adj_ro, wts_ro) on the
 * induced subgraph (heuristics.py:13-35) without re-slicing the adjacency.  Arrays are instance-major:
 * init_state/state [num_instances][num_nodes], rounds/totals [num_instances][num_graphs];
 * prio [num_nodes] shared by all instances (prio_stride = 0) or one slice per instance.
 * state keeps the given init_state values for the vertices that were taken out. */
int dgcn_lgs_masked_batch(const DgcnBatch* batch, const double* prio, int64_t prio_stride, const uint8_t* init_state,
                          int32_t num_instances, int32_t max_rounds, uint8_t* state, int32_t* rounds,
                          const double* sum_weights, double* totals, int32_t* status, void* stream);

/* ---- A1-A10 in one launch: mwis_gdpg_call.py:200-235 solve_mwis for a whole batch ---------------
 * adjacency + vertex weights in, membership out; one workgroup keeps one graph in LDS from support
 * construction (gcn/utils.py:120-127) through every layer (gcn/models.py:536-573) to the greedy
 * rounds (heuristics.py:77-116).  Same arithmetic contract as the separate entry points, so scores
 * and sets are bit-identical to supports -> forward(mode 0) -> lgs.
 * Handles F->c->..->c->1 layer stacks with hidden widths c <= 32 (narrower than 32: computed 32 wide with
 * zero weights, which leaves the real features' bits unchanged) on graphs of <= 512 vertices whose image (with the entry values
 * in LDS or, for larger graphs, in the global scratch) fits the LDS;
 * dgcn_solve_supported() tells (1/0) so the caller can route other shapes through the separate calls.
 * scores (float[num_nodes]), rounds, totals may be NULL.  weights NULL or predict_mwis = 0: the
 * priority is the score itself (mwis_dqn_call.py:234). */
int dgcn_solve_supported(const DgcnBatch* batch, const DgcnModel* model);
/* Graphs beyond that (the multi-channel joint conflict graphs of wireless_dqn_test_mc.py:161 have K * nflows vertices;
 * ER(500, 0.1) already has more entries than one CU's LDS holds) and layer stacks wider than 32 take the ANY-SIZE path
 * inside the same two entry points.  One launch where a kernel takes the shape: one- and two-layer models (the multi-channel
 * launchers' own --num_layer=1, bash/twc_major_wireless_mc_test.sh:3) up to 9 600 vertices, deep c32 stacks up to 976 vertices
 * with any number of entries and up to 1 920 vertices beyond that - a residual step likewise (the residual graph's support is
 * formed from the adjacency and the running state inside the launch).  Otherwise the residual graph is re-sliced on the device
 * (renumbered, with its own support), the forward pass runs layer by layer (mode 0's kernels: same bits), the greedy step in
 * kernels of its own.  Nothing returns to the host, nothing changes in the results.  dgcn_solve_path: 1 = fused kernels, 2 = any-size path (graphs
 * up to 9 600 vertices, [I, L] models with one output per vertex), 0 = neither (DGCN_ERR_UNSUPPORTED).
 * dgcn_set_general(1) sends every shape down the any-size path (tests, A/B runs); -1 = automatic (default).
 * Same word as dgcn_set_option("general", ..). */
int dgcn_solve_path(const DgcnBatch* batch, const DgcnModel* model);
void dgcn_set_general(int32_t setting);
int32_t dgcn_get_general(void);
/* The several-workgroups-per-graph launch variant (small batches): -1 = chosen automatically (default), 0 = off,
 * K >= 2 = forced.  Process-wide, atomic; same word as dgcn_set_option("fused_cluster", ..).  The
 * library's own recovery from DGCN_FAULT_CLUSTER calls dgcn_set_cluster(0). */
void dgcn_set_cluster(int32_t setting);
int32_t dgcn_get_cluster(void);
/* Bytes of device scratch dgcn_solve_batch / dgcn_solve_residual_batch need for this batch.  Graphs whose
 * whole image fits the LDS need a token amount; larger ones (e.g. 500 vertices, 5 000 edges) keep their
 * entry values in this scratch (one float per entry slot) and only states + gather words in LDS; the any-size
 * path keeps the re-sliced batch, its support, the layer-by-layer buffers and up to 64 rollout instances here. */
size_t dgcn_solve_workspace(const DgcnBatch* batch, const DgcnModel* model);
int dgcn_solve_batch(const DgcnBatch* batch, const DgcnModel* model, const double* dinv_table, int32_t table_len,
                     const float* X, float x_const, const double* weights, int32_t predict_mwis,
                     float* scores, uint8_t* state, int32_t* rounds, double* totals, int32_t* status,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- F1/F2: one step of the iterative solvers on the RESIDUAL graph, batched, in one launch ------
 * mwis_gdpg_call.py:278-318 (solve_mwis_dit), :343-384 (solve_mwis_cit), :596-659 (solve_mwis_rollout)
 * re-slice the SciPy matrix to the undecided vertices before every GCN pass.  Here `state` (uint8 per
 * vertex, in/out) plays nIS_vec: 0 = undecided (in the residual graph), 1 = in the set, 2 = excluded;
 * vertices with state != 0 are dropped while the LDS image is built, so the forward pass runs on the
 * induced subgraph with its own degrees, exactly as the re-sliced matrix would.
 *   greedy_mode 0: `max_rounds` local-greedy rounds by priority (0 = until all decided); dit uses 1
 *   greedy_mode 1: the best-priority vertex joins (np.argmax: lowest index among equals)
 *   greedy_mode 2: the first `beam` (<= 64) vertices by priority are candidates; each is completed by
 *                  a local greedy search by weight on the residual graph minus its closed
 *                  neighbourhood; the candidate with the largest weight + completion total joins
 *                  (totals within 1e-12 relative count as tied: first candidate wins)
 * feature_mode 0: X / x_const as in dgcn_solve_batch (X rows of removed vertices are ignored);
 * feature_mode 1: every feature of vertex v is (float)(w[v] / (max residual w + 1e-9)) (X must be NULL).
 * A graph with no undecided vertex, or no positive weight left (np.sum(wts_nn) <= 0 -> break), is left
 * untouched; otherwise *progress += 1.  The caller repeats the launch until *progress stays 0.
 * options: DGCN_RESIDUAL_* bits (the rollout variants mwis_gdpg_call.py:413-594).
 * rounds[g]: rounds run by this launch; totals[g]: weight (or priority) of the vertices that joined in
 * THIS launch; scores: residual-graph scores (0 for removed vertices).  Same shapes as
 * dgcn_solve_batch (dgcn_solve_path). */
#define DGCN_RESIDUAL_SCORES_GIVEN 1        /* `scores` is an INPUT (one forward pass on the full graph, done by the
                                              caller): no forward pass here - solve_mwis_rollout00 / rollout0 */
#define DGCN_RESIDUAL_COMPLETE_BY_PRIORITY 2 /* rollout completions ordered by priority instead of weight
                                              (greedy_search(adj_ro, gw_ro)): solve_mwis_rollout0 / rollout1 */
#define DGCN_RESIDUAL_FINISH_SMALL 4         /* after this call's step, a graph with at most 64 undecided vertices runs ALL its
                                              remaining steps inside the same call (csrc/tail.hip: one more launch, the
                                              residual graph held on the chip); graphs with more go on step by step, call
                                              by call.  Final states as without the bit; rounds[g] / totals[g] then are the
                                              sums over every step the call ran for graph g (rounds: + 1 per step of the
                                              tail), scores of a graph finished this way read 0 (as after the last step).
                                              Taken for [I, L] stacks F -> 32 -> ... -> 32 -> 1 of >= 3 layers with X == NULL
                                              and without DGCN_RESIDUAL_SCORES_GIVEN; ignored otherwise */
int dgcn_solve_residual_batch(const DgcnBatch* batch, const DgcnModel* model, const double* dinv_table,
                              int32_t table_len, const float* X, float x_const, int32_t feature_mode,
                              const double* weights, int32_t predict_mwis, int32_t greedy_mode, int32_t max_rounds,
                              int32_t beam, int32_t options, float* scores, uint8_t* state, int32_t* rounds, double* totals,
                              int32_t* progress, int32_t* status, void* workspace, size_t workspace_bytes,
                              void* stream);

/* ---- host to host: the reference's call pattern (one graph, or one directory of graphs, per call) -----------
 * mwis_dqn_call.py:140-143 / mwis_gdpg_call.py:211-216 run one graph per sess.run; mwis_dqn_test.py:304-321 walks a
 * directory.  A DgcnHostSolver keeps `depth` slots of pinned staging memory, device buffers, a stream and an event:
 *   submit: dgcn_pack_batch into the slot's pinned memory -> one host-to-device copy -> dgcn_solve_batch -> one
 *           device-to-host copy, all asynchronous; returns the slot index (>= 0) or a DGCN_ERR_* code (< 0)
 *           (a packed batch of at most option "host_direct_bytes", default 2 MB, is not copied: the kernel reads the pinned
 *           staging memory and writes the pinned result memory itself)
 *   result: waits for that slot; hands out pointers into its pinned result memory (valid until the slot's next submit):
 *           state[num_nodes] (0 undecided / 1 in the set / 2 excluded), totals[num_graphs], rounds[num_graphs],
 *           scores[num_nodes] when created with want_scores
 * Up to `depth` batches may be in flight; slots are used round-robin and a slot must be read before it is re-used.
 * `model` descriptors are copied, the DEVICE weights they point at and `dinv_table` (DEVICE, float64 d^-1/2) must outlive
 * the object.  All graph pointers are HOST pointers as in dgcn_pack_batch; nothing is kept after submit returns.
 * Shapes outside the fused kernel: DGCN_ERR_UNSUPPORTED (use the separate calls).  One thread at a time per object.
 * model == NULL: no GCN - the plain local greedy search with the weights as priorities (heuristics.py:77-116,
 * local_greedy_search / greedy_search; dgcn_lgs_batch underneath, any graph size); dinv_table is ignored, weights are
 * required and an adjacency with a self-loop is refused at submit (the reference never terminates on one). */
typedef struct DgcnHostSolver DgcnHostSolver;
int dgcn_host_solver_create(const DgcnModel* model, const double* dinv_table, int32_t table_len, int32_t predict_mwis,
                            float x_const, int32_t want_scores, int32_t depth, int32_t pack_threads, DgcnHostSolver** out);
void dgcn_host_solver_destroy(DgcnHostSolver* solver);
int dgcn_host_solver_submit(DgcnHostSolver* solver, const void* const* indptr_host, const void* const* indices_host,
                            const double* const* weights_host, const int32_t* num_nodes_host, int32_t num_graphs,
                            int32_t index_bytes);
int dgcn_host_solver_result(DgcnHostSolver* solver, int32_t slot, const uint8_t** state, const double** totals,
                            const int32_t** rounds, const float** scores /* NULL unless created with want_scores */,
                            int32_t* status_bits, int32_t* num_nodes, int32_t* num_graphs);

/* ---- per-kernel timing for bench.py's roofline line (HIP events on the launch stream) ---------
 * enable(1) makes every launch of the named kernel families record an event pair (enable(N), N > 1: every N-th launch only -
 * an A/B switch for what the instrumentation costs; 0: off);
 * read() synchronises those events and returns the summed milliseconds and launch count. */
int dgcn_timing_enable(int32_t on);
int dgcn_timing_reset(void);
int dgcn_timing_read(const char* kernel, double* total_ms, int64_t* launches);
/* N = every N-th launch carries (carried, once timing is off again) an event pair: read()'s totals cover the sampled launches
 * only, a caller that wants per-step figures multiplies both by N.  Event pairs are kept per device: a launch takes its pair
 * from the device that is current when it is issued. */
int32_t dgcn_timing_sampling(void);

#ifdef __cplusplus
}
#endif
#endif /* DGCN_H */
