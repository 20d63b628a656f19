// Support construction on device: L = I - D^-1/2 A D^-1/2 for a block-diagonal batch.
// Replaces gcn/utils.py:120-127 (normalize_adj) + :258-274 (simple_polynomials, k = 1), which the
// reference runs per graph in SciPy float64 (~2 ms/graph) before every forward pass.
//
// HBM-bound, runs once per batch: reads row_ptr/col_idx once, writes (col, val) once.
// Layout: 8 lanes per row, so a wave writes 8 rows' entries with 32-byte segments per row.
#include "common.h"

namespace dgcn {

constexpr int kSupRowsPerBlock = 32;  // 256 threads / 8 lanes per row

__global__ __launch_bounds__(256) void k_supports(const int32_t* __restrict__ graph_ptr,
                                                  const int32_t* __restrict__ row_ptr,
                                                  const int32_t* __restrict__ col_idx, int num_nodes, int tiles,
                                                  const double* __restrict__ dinv_table, int table_len,
                                                  int32_t* __restrict__ lap_row_ptr, int32_t* __restrict__ lap_col,
                                                  float* __restrict__ lap_val, int32_t* __restrict__ status) {
    const int g = blockIdx.x / tiles;
    const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1];
    const int sub = threadIdx.x & 7;
    const int v = n0 + (blockIdx.x % tiles) * kSupRowsPerBlock + (threadIdx.x >> 3);
    if (v >= n1) return;
    const int rs = row_ptr[v], re = row_ptr[v + 1];
    const int deg = re - rs;
    int fault = 0;
    double dv = 0.0;
    if (deg < table_len) dv = dinv_table[deg]; else fault |= DGCN_FAULT_DEGREE_RANGE;
    const int out = rs + v;  // one extra (diagonal) entry per preceding row
    if (sub == 0) {
        lap_row_ptr[v] = out;
        lap_col[out] = v;
        lap_val[out] = 1.0f;  // (I - A_hat)[v][v] with a zero-diagonal adjacency
        if (v == num_nodes - 1) lap_row_ptr[num_nodes] = re + num_nodes;
    }
    for (int j = rs + sub; j < re; j += 8) {
        const int u = col_idx[j];
        float val = 0.0f;
        if (u < n0 || u >= n1) {
            fault |= DGCN_FAULT_BAD_COLUMN;
        } else {
            if (u == v) fault |= DGCN_FAULT_SELF_LOOP;
            const int du = row_ptr[u + 1] - row_ptr[u];
            double d = 0.0;
            if (du < table_len) d = dinv_table[du]; else fault |= DGCN_FAULT_DEGREE_RANGE;
            // reference order: (A_vu * dinv[u]) * dinv[v] in float64, negated by "eye - A_hat",
            // then TF's float64 -> float32 feed cast
            val = (float)(-(d * dv));
        }
        lap_col[j + v + 1] = u;
        lap_val[j + v + 1] = val;
    }
    if (fault) atomicOr(status, fault);
}

}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_supports_batch(const DgcnBatch* b, const double* dinv_table, int32_t table_len,
                                   int32_t* lap_row_ptr, int32_t* lap_col, float* lap_val,
                                   int32_t* status, void* stream) {
    if (!b || !dinv_table || !lap_row_ptr || !lap_col || !lap_val || !status)
        return fail(DGCN_ERR_ARG, "dgcn_supports_batch: null argument");
    if (b->num_graphs <= 0 || b->num_nodes <= 0) return DGCN_OK;
    if (table_len <= 0 || b->max_nodes <= 0) return fail(DGCN_ERR_ARG, "dgcn_supports_batch: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int tiles = ceil_div(b->max_nodes, kSupRowsPerBlock);
    dim3 grid((unsigned)tiles * (unsigned)b->num_graphs);
    TimedLaunch t("supports", s);
    DGCN_LAUNCH(t, k_supports, grid, dim3(256), 0, s, b->graph_ptr, b->row_ptr, b->col_idx, b->num_nodes,
                       tiles, dinv_table, table_len, lap_row_ptr, lap_col, lap_val, status);
    return check_launch("k_supports");
}
