// Second-order Chebyshev support on device: T_2 = L . L for a block-diagonal batch, formed EXPLICITLY as
// the reference does (gcn/utils.py:268-271: "t_new = t_k[-1]*laplacian" is SciPy's csr_matmat on float64
// CSR matrices with sorted columns), so that the float32 values TensorFlow is fed are reproduced bit for bit:
//   * for output row i the partial products are added in the order of j in row i of L (columns ascending,
//     the diagonal in its sorted place): sums[k] = sums[k] + L[i,j] * L[j,k], float64, multiply and add
//     rounded separately (SciPy's C++ is compiled without FMA contraction on x86-64);
//   * entries whose sum is exactly zero are dropped (csr_matmat: "if (sums[head] != 0)");
//   * the result is cast float64 -> float32 at the TF feed.
// Output rows are stored with ascending columns, global vertex ids.
//
// One wave owns one output row at a time: a dense float64 accumulator of the graph's N_g columns in LDS
// (the rows of L^2 of an ER N=200 p=0.1 graph are ~87 % dense), lanes spread over the entries of row j of
// L, j sequential.  Two passes with identical arithmetic: COUNT (row lengths -> exclusive scan = row_ptr)
// and FILL - the caller allocates the arrays in between (include/dgcn.h).  Runs once per batch; the
// adjacency rows must be sorted by column (the host packer guarantees it), otherwise values still agree
// to float64 rounding but no longer bit for bit.
#include "common.h"

namespace dgcn {

constexpr int kS2RowsPerTile = 32;

template <bool FILL>
__global__ __launch_bounds__(256) void k_supports2(const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ row_ptr,
                                                   const int32_t* __restrict__ col_idx, int tiles, int stride,
                                                   const double* __restrict__ dinv_table, int table_len,
                                                   int32_t* __restrict__ row_ptr2, int32_t* __restrict__ col2,
                                                   float* __restrict__ val2, int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double s2_lds[];
    const int g = blockIdx.x / tiles;
    const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1];
    const int ng = n1 - n0;
    const int r0 = (blockIdx.x % tiles) * kS2RowsPerTile;
    if (r0 >= ng) return;
    const int r1 = min(r0 + kS2RowsPerTile, ng);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    double* dv = s2_lds;                                // [stride] d^-1/2 of every vertex of the graph
    double* acc = s2_lds + (size_t)(1 + wave) * stride;  // this wave's accumulator row
    int fault = 0;
    for (int k = threadIdx.x; k < ng; k += blockDim.x) {
        const int deg = row_ptr[n0 + k + 1] - row_ptr[n0 + k];
        double d = 0.0;
        if (deg < table_len) d = dinv_table[deg]; else fault |= DGCN_FAULT_DEGREE_RANGE;
        dv[k] = d;
    }
    __syncthreads();
    for (int i = r0 + wave; i < r1; i += waves) {
        for (int k = lane; k < ng; k += 64) acc[k] = 0.0;
        const int rs = row_ptr[n0 + i], re = row_ptr[n0 + i + 1];
        const double di = dv[i];
        bool diag_done = false;
        int p = rs;
        while (p < re || !diag_done) {
            // next column j of row i of L in ascending order (adjacency columns + the diagonal)
            int j;
            double lij;
            const int cj = p < re ? col_idx[p] - n0 : 0x7fffffff;
            if (!diag_done && cj > i) {
                j = i;
                lij = 1.0;
                diag_done = true;
            } else {
                ++p;
                if (cj < 0 || cj >= ng) { fault |= DGCN_FAULT_BAD_COLUMN; continue; }
                if (cj == i) { fault |= DGCN_FAULT_SELF_LOOP; continue; }
                j = cj;
                lij = -(dv[j] * di);
            }
            // row j of L: its adjacency entries, then the diagonal (distinct columns: any lane order)
            const int js = row_ptr[n0 + j], je = row_ptr[n0 + j + 1];
            const double dj = dv[j];
            __builtin_amdgcn_wave_barrier();
            for (int t = lane; t <= je - js; t += 64) {
                int k;
                double ljk;
                if (t < je - js) {
                    k = col_idx[js + t] - n0;
                    if (k < 0 || k >= ng || k == j) continue;  // reported when row j is the output row
                    ljk = -(dj * dv[k]);
                } else {
                    k = j;
                    ljk = 1.0;
                }
                acc[k] = __dadd_rn(acc[k], __dmul_rn(lij, ljk));
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this step's LDS updates precede the next step's reads
        }
        int cnt = 0;
        const int start = FILL ? row_ptr2[n0 + i] : 0;
        for (int kb = 0; kb < ng; kb += 64) {
            const int k = kb + lane;
            const double a = k < ng ? acc[k] : 0.0;
            const bool nz = a != 0.0;
            const unsigned long long m = __ballot(nz);
            if (FILL && nz) {
                const int pos = start + cnt + __popcll(m & ((1ull << lane) - 1ull));
                col2[pos] = n0 + k;
                val2[pos] = (float)a;
            }
            cnt += __popcll(m);
        }
        if (!FILL && lane == 0) row_ptr2[n0 + i] = cnt;
        __builtin_amdgcn_wave_barrier();
    }
    if (fault) atomicOr(status, fault);
}

// Exclusive prefix sum of a[0..n) in place, a[n] = total.  One workgroup (the entry point has no scratch for a multi-block
// scan), sixteen consecutive counts per thread and pass - 16 384 per pass: the 230 400 rows of 256 joint 3 x 300 graphs take
// 15 passes (a count per thread and pass, round 5: 225 passes, 244 us of a 2.3 ms [I, L, L.L] solve).
constexpr int kScanPer = 16;
__global__ __launch_bounds__(1024) void k_scan_counts(int32_t* __restrict__ a, int n) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024 * kScanPer) {
        const int i0 = base + threadIdx.x * kScanPer;
        int x[kScanPer];
        if (i0 + kScanPer <= n && (reinterpret_cast<uintptr_t>(a + i0) & 15) == 0) {
#pragma unroll
            for (int q = 0; q < kScanPer / 4; ++q) {
                const int4 v = *reinterpret_cast<const int4*>(a + i0 + 4 * q);
                x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < kScanPer; ++q) x[q] = i0 + q < n ? a[i0 + q] : 0;
        }
        int mine = 0;
#pragma unroll
        for (int q = 0; q < kScanPer; ++q) mine += x[q];
        int incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        int run = before + incl - mine;  // everything in front of this thread's sixteen
#pragma unroll
        for (int q = 0; q < kScanPer; ++q) {
            const int v = x[q];
            if (i0 + q < n) a[i0 + q] = run;
            run += v;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) a[n] = carry_s;
}

static int supports2_geometry(const DgcnBatch* b, int* stride, int* waves, size_t* lds) {
    *stride = (max(b->max_nodes, 1) + 1) & ~1;
    for (int w = 4; w >= 1; w >>= 1) {
        const size_t need = (size_t)(1 + w) * *stride * sizeof(double);
        if (need <= 150 * 1024) { *waves = w; *lds = need; return DGCN_OK; }
    }
    return fail(DGCN_ERR_UNSUPPORTED, "dgcn_supports2: graphs of %d vertices exceed the LDS accumulator (max 9600)", b->max_nodes);
}

template <bool FILL>
static int supports2_launch(const DgcnBatch* b, const double* dinv_table, int table_len, int32_t* row_ptr2, int32_t* col2,
                            float* val2, int32_t* status, hipStream_t s) {
    int stride, waves;
    size_t lds;
    int rc = supports2_geometry(b, &stride, &waves, &lds);
    if (rc) return rc;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_supports2<FILL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DGCN_ERR_LAUNCH, "k_supports2: cannot reserve %zu bytes of LDS", lds);
    }
    const int tiles = ceil_div(b->max_nodes, kS2RowsPerTile);
    TimedLaunch t(FILL ? "supports2_fill" : "supports2_count", s);
    DGCN_LAUNCH(t, k_supports2<FILL>, dim3((unsigned)tiles * (unsigned)b->num_graphs), dim3(64 * waves), lds, s, b->graph_ptr,
                b->row_ptr, b->col_idx, tiles, stride, dinv_table, table_len, row_ptr2, col2, val2, status);
    return check_launch("k_supports2");
}

}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_supports2_count_batch(const DgcnBatch* b, const double* dinv_table, int32_t table_len,
                                          int32_t* lap2_row_ptr, int32_t* status, void* stream) {
    if (!b || !dinv_table || !lap2_row_ptr || !status) return fail(DGCN_ERR_ARG, "dgcn_supports2_count_batch: null argument");
    if (table_len <= 0) return fail(DGCN_ERR_ARG, "dgcn_supports2_count_batch: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    if (b->num_graphs <= 0 || b->num_nodes <= 0) {
        if (hipMemsetAsync(lap2_row_ptr, 0, sizeof(int32_t), s) != hipSuccess) return fail(DGCN_ERR_LAUNCH, "memset");
        return DGCN_OK;
    }
    int rc = supports2_launch<false>(b, dinv_table, table_len, lap2_row_ptr, nullptr, nullptr, status, s);
    if (rc) return rc;
    TimedLaunch t("supports2_scan", s);
    DGCN_LAUNCH(t, k_scan_counts, dim3(1), dim3(1024), 0, s, lap2_row_ptr, b->num_nodes);
    return check_launch("k_scan_counts");
}

extern "C" int dgcn_supports2_fill_batch(const DgcnBatch* b, const double* dinv_table, int32_t table_len,
                                         const int32_t* lap2_row_ptr, int32_t* lap2_col, float* lap2_val,
                                         int32_t* status, void* stream) {
    if (!b || !dinv_table || !lap2_row_ptr || !lap2_col || !lap2_val || !status)
        return fail(DGCN_ERR_ARG, "dgcn_supports2_fill_batch: null argument");
    if (b->num_graphs <= 0 || b->num_nodes <= 0) return DGCN_OK;
    return supports2_launch<true>(b, dinv_table, table_len, const_cast<int32_t*>(lap2_row_ptr), lap2_col, lap2_val, status,
                                  (hipStream_t)stream);
}
