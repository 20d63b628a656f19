/*
 * f32_orders.c - TEST INFRASTRUCTURE ONLY.  The reference's GraphConvolution stack (gcn/layers.py:189-216,
 * gcn/models.py:536-573) evaluated in float32 under SEVERAL summation orders.
 *
 * Why: the reference's float32 arithmetic happens inside TensorFlow (sparse_tensor_dense_matmul, matmul, add_n:
 * gcn/layers.py:29-31, 206, 208), which is not installable here - its summation order is pinned by nothing.  What CAN be
 * done is to bound it: every plausible float32 implementation of the same formula is one of a family that differs only in
 * (a) fused multiply-add or separate multiply and add, (b) the order in which a row's nonzeros are visited (COO storage
 * order = ascending column, or diagonal first as the HIP kernels store L), (c) how the k loop of the dense product is
 * split (one chain; blocks of 8 or 16 added in order - split-k GEMMs; a pairwise tree).  tools/f32_envelope.py evaluates
 * the BASELINE configurations under all of them and reports the spread, each one's distance from the float64 evaluation,
 * and whether any of them changes a selected set.
 *
 *   spmm_fma  0: acc = acc + v * z (two roundings, TF's CPU SparseTensorDenseMatMul without contraction)   1: fmaf
 *   mm_mode   0: one k chain, multiply and add rounded separately      1: one k chain of fmaf (Eigen gebp with FMA)
 *             2 / 3: fmaf chains over blocks of 8 / 16 k, block sums added in ascending order (split-k)
 *             4: pairwise tree over k (halves), products rounded, then adds
 * The entries of a row are visited in the order the caller stored them.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static float actf(float x, int act) {
    if (act == 1) return x > 0.0f ? x : 0.2f * x;
    if (act == 2) return x > 0.0f ? x : 0.0f;
    return x;
}

static float tree_sum(const float* p, int n) {
    if (n == 1) return p[0];
    const int h = n / 2;
    return tree_sum(p, h) + tree_sum(p + h, n - h);
}

static float dot_mode(const float* h, int ldh_unused, const float* W, int ctot, int n, int cin, int mode, float h_const, int has_h) {
    (void)ldh_unused;
    float acc = 0.0f;
    if (mode == 0) {
        for (int k = 0; k < cin; ++k) { const float pr = (has_h ? h[k] : h_const) * W[k * ctot + n]; acc = acc + pr; }
        return acc;
    }
    if (mode == 1) {
        for (int k = 0; k < cin; ++k) acc = fmaf(has_h ? h[k] : h_const, W[k * ctot + n], acc);
        return acc;
    }
    if (mode == 2 || mode == 3) {
        const int blk = mode == 2 ? 8 : 16;
        for (int k0 = 0; k0 < cin; k0 += blk) {
            float part = 0.0f;
            for (int k = k0; k < cin && k < k0 + blk; ++k) part = fmaf(has_h ? h[k] : h_const, W[k * ctot + n], part);
            acc = k0 == 0 ? part : acc + part;
        }
        return acc;
    }
    {
        float pr[512];
        const int m = cin > 512 ? 512 : cin;
        for (int k = 0; k < m; ++k) pr[k] = (has_h ? h[k] : h_const) * W[k * ctot + n];
        return tree_sum(pr, m);
    }
}

/* dims[l], dims[l+1]: in / out widths; weights[l]: [in][2*out] (W0 | W1); biases[l] or NULL; acts[l].
 * (row_ptr, col, val): the support L in CSR, entries in the order they are to be visited. */
int ord_forward(int num_nodes, const int32_t* row_ptr, const int32_t* col, const float* val, int num_layers,
                const int32_t* dims, const float* const* weights, const float* const* biases, const int32_t* acts,
                const float* X, float x_const, int spmm_fma, int mm_mode, float* scores) {
    int maxd = 0;
    for (int l = 0; l <= num_layers; ++l) if (dims[l] > maxd) maxd = dims[l];
    float* Z = (float*)malloc((size_t)num_nodes * 2 * maxd * sizeof(float) + 16);
    float* Ha = (float*)malloc((size_t)num_nodes * maxd * sizeof(float) + 16);
    float* Hb = (float*)malloc((size_t)num_nodes * maxd * sizeof(float) + 16);
    if (!Z || !Ha || !Hb) { free(Z); free(Ha); free(Hb); return -1; }
    const float* H = X;
    int ldh = dims[0];
    for (int l = 0; l < num_layers; ++l) {
        const int cin = dims[l], cout = dims[l + 1], ctot = 2 * cout;
        for (int r = 0; r < num_nodes; ++r)
            for (int n = 0; n < ctot; ++n)
                Z[(size_t)r * ctot + n] = dot_mode(H ? H + (size_t)r * ldh : NULL, ldh, weights[l], ctot, n, cin, mm_mode, x_const, H != NULL);
        float* out = (l == num_layers - 1) ? scores : (H == Ha ? Hb : Ha);
        for (int v = 0; v < num_nodes; ++v)
            for (int c = 0; c < cout; ++c) {
                float acc = 0.0f;
                for (int j = row_ptr[v]; j < row_ptr[v + 1]; ++j) {
                    const float z = Z[(size_t)col[j] * ctot + cout + c];
                    if (spmm_fma) acc = fmaf(val[j], z, acc);
                    else { const float pr = val[j] * z; acc = acc + pr; }
                }
                float o = Z[(size_t)v * ctot + c] + acc;   /* tf.add_n([support_0 . pre_0, support_1 . pre_1]) */
                if (biases[l]) o = o + biases[l][c];
                out[(size_t)v * cout + c] = actf(o, acts[l]);
            }
        H = out;
        ldh = cout;
    }
    free(Z);
    free(Ha);
    free(Hb);
    return 0;
}
