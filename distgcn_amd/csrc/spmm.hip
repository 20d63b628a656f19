// Batched block-diagonal CSR SpMM with a fused GraphConvolution epilogue.
//   Y[v,:] = act( Y0[v,:] + sum_j S.val[j] * Z[S.col[j],:] + bias )
// Replaces tf sparse_tensor_dense_matmul + add_n + bias + activation of the reference's
// GraphConvolution._call (gcn/layers.py:206-216).
//
// Roofline: HBM-bound.  Algorithmic bytes per launch:
//   sum_g [ nnz_g*(4+4) + (N_g+1)*4 ] + 2*4*C*sum_g N_g   (+ 4*C*sum N_g when Y0 is given)
// Design for gfx950:
//   * LPR lanes own one output row; each lane keeps VEC consecutive features in registers, so a
//     neighbour row of Z (C=32 floats = 128 B) is fetched by 8 lanes x 16 B = one full line and
//     the row sum needs no cross-lane reduction.  The sum is a sequential fmaf chain in CSR
//     order: deterministic, and mirrored by oracle/dgcn_oracle.c.
//   * "lds" variant: a workgroup serves a tile of rows of ONE graph; it first copies that graph's
//     slice of Z (N_g x C floats, 25.6 KB at N=200,C=32) into LDS with coalesced 16-byte loads, so
//     every neighbour gather is a ds_read_b128 instead of an L1/L2 round trip, and (when it fits)
//     the tile's (col,val) range too, so the per-row index walk is LDS-resident as well.
//   * "global" variant: same row loop gathering straight from L2 - used when a graph's slice does
//     not fit the LDS budget, or when the caller passes no graph_ptr.
#include "common.h"

namespace dgcn {

template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<1> { using type = float; };

__device__ __forceinline__ float4 vfma(float a, float4 z, float4 acc) {
    acc.x = fmaf(a, z.x, acc.x); acc.y = fmaf(a, z.y, acc.y);
    acc.z = fmaf(a, z.z, acc.z); acc.w = fmaf(a, z.w, acc.w);
    return acc;
}
__device__ __forceinline__ float vfma(float a, float z, float acc) { return fmaf(a, z, acc); }
__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }
__device__ __forceinline__ float4 vact(float4 a, int act) {
    return make_float4(apply_act(a.x, act), apply_act(a.y, act), apply_act(a.z, act), apply_act(a.w, act));
}
__device__ __forceinline__ float vact(float a, int act) { return apply_act(a, act); }
template <int VEC> __device__ __forceinline__ typename VecT<VEC>::type vzero();
template <> __device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float vzero<1>() { return 0.f; }

// One output row: sequential fmaf chain over [rs, re) in CSR order.
// cols/vals are indexed with (j - cbase); zrow(u) returns the address of Z's row for column id u.
template <int VEC, typename ColP, typename ValP, typename ZRow>
__device__ __forceinline__ typename VecT<VEC>::type row_sum(ColP cols, ValP vals, int rs, int re, int foff,
                                                            ZRow zrow) {
    using V = typename VecT<VEC>::type;
    V acc = vzero<VEC>();
    int j = rs;
    for (; j + 4 <= re; j += 4) {
        const int c0 = cols[j], c1 = cols[j + 1], c2 = cols[j + 2], c3 = cols[j + 3];
        const float a0 = vals[j], a1 = vals[j + 1], a2 = vals[j + 2], a3 = vals[j + 3];
        const V z0 = *reinterpret_cast<const V*>(zrow(c0) + foff);
        const V z1 = *reinterpret_cast<const V*>(zrow(c1) + foff);
        const V z2 = *reinterpret_cast<const V*>(zrow(c2) + foff);
        const V z3 = *reinterpret_cast<const V*>(zrow(c3) + foff);
        acc = vfma(a0, z0, acc);
        acc = vfma(a1, z1, acc);
        acc = vfma(a2, z2, acc);
        acc = vfma(a3, z3, acc);
    }
    for (; j < re; ++j) {
        const V z = *reinterpret_cast<const V*>(zrow(cols[j]) + foff);
        acc = vfma(vals[j], z, acc);
    }
    return acc;
}

template <int VEC>
__device__ __forceinline__ void epilogue_store(typename VecT<VEC>::type acc, int v, int foff, const float* Y0,
                                               int ldy0, const float* bias, int act, float* Y, int ldy) {
    using V = typename VecT<VEC>::type;
    if (Y0) acc = vadd(*reinterpret_cast<const V*>(Y0 + (size_t)v * ldy0 + foff), acc);
    if (bias) acc = vadd(acc, *reinterpret_cast<const V*>(bias + foff));
    *reinterpret_cast<V*>(Y + (size_t)v * ldy + foff) = vact(acc, act);
}

// ---------------------------------------------------------------------------------------------
// LDS variant: blockIdx.x = graph * tiles + tile
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void k_spmm_lds(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                  const float* __restrict__ values, const int32_t* __restrict__ graph_ptr,
                                                  int tiles, int rows_per_tile, const float* __restrict__ Z, int ldz,
                                                  int C, int zs, int csr_cap, const float* __restrict__ Y0, int ldy0,
                                                  const float* __restrict__ bias, int act, float* __restrict__ Y,
                                                  int ldy) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = blockIdx.x / tiles;
    const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1];
    const int r0 = n0 + (blockIdx.x % tiles) * rows_per_tile;
    if (r0 >= n1) return;
    const int r1 = min(r0 + rows_per_tile, n1);
    const int ng = n1 - n0;
    float* zsm = smem;
    // ---- stage this graph's slice of Z: coalesced VEC-wide loads, LDS row stride zs
    {
        using V = typename VecT<VEC>::type;
        const int per_row = C / VEC;
        const int total = ng * per_row;
        for (int i = threadIdx.x; i < total; i += 256) {
            const int row = i / per_row, q = i - row * per_row;
            *reinterpret_cast<V*>(zsm + row * zs + q * VEC) =
                *reinterpret_cast<const V*>(Z + (size_t)(n0 + row) * ldz + q * VEC);
        }
    }
    const int e0 = row_ptr[r0], e1 = row_ptr[r1];
    const bool csr_in_lds = (e1 - e0) <= csr_cap;
    int32_t* csm = reinterpret_cast<int32_t*>(smem + (size_t)((ng * zs + 3) & ~3));
    float* vsm = reinterpret_cast<float*>(csm + csr_cap);
    if (csr_in_lds) {
        for (int i = e0 + threadIdx.x; i < e1; i += 256) {
            csm[i - e0] = col_idx[i] - n0;  // local ids
            vsm[i - e0] = values[i];
        }
    }
    __syncthreads();
    const int grp = threadIdx.x / LPR, sub = threadIdx.x % LPR;
    const int foff = sub * VEC;
    if (foff >= C) return;
    constexpr int kGroups = 256 / LPR;
    auto zrow_local = [&](int u) -> const float* { return zsm + u * zs; };
    for (int v = r0 + grp; v < r1; v += kGroups) {
        const int rs = row_ptr[v], re = row_ptr[v + 1];
        typename VecT<VEC>::type acc;
        if (csr_in_lds) {
            acc = row_sum<VEC>(csm - e0, vsm - e0, rs, re, foff, zrow_local);
        } else {
            auto zrow_glob = [&](int u) -> const float* { return zsm + (u - n0) * zs; };
            acc = row_sum<VEC>(col_idx, values, rs, re, foff, zrow_glob);
        }
        epilogue_store<VEC>(acc, v, foff, Y0, ldy0, bias, act, Y, ldy);
    }
}

// ---------------------------------------------------------------------------------------------
// Global-gather variant: no graph structure needed.
template <int VEC, int LPR>
__global__ __launch_bounds__(256) void k_spmm_global(const int32_t* __restrict__ row_ptr,
                                                     const int32_t* __restrict__ col_idx,
                                                     const float* __restrict__ values, int num_rows,
                                                     const float* __restrict__ Z, int ldz, int C,
                                                     const float* __restrict__ Y0, int ldy0,
                                                     const float* __restrict__ bias, int act,
                                                     float* __restrict__ Y, int ldy) {
    constexpr int kGroups = 256 / LPR;
    const int grp = threadIdx.x / LPR, sub = threadIdx.x % LPR;
    const int foff = sub * VEC;
    if (foff >= C) return;
    auto zrow = [&](int u) -> const float* { return Z + (size_t)u * ldz; };
    for (int v = blockIdx.x * kGroups + grp; v < num_rows; v += gridDim.x * kGroups) {
        const int rs = row_ptr[v], re = row_ptr[v + 1];
        auto acc = row_sum<VEC>(col_idx, values, rs, re, foff, zrow);
        epilogue_store<VEC>(acc, v, foff, Y0, ldy0, bias, act, Y, ldy);
    }
}

static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    return atoi(e);
}

template <int VEC, int LPR>
static int launch_spmm(const DgcnCsr* S, const int32_t* graph_ptr, int B, int max_nodes, const float* Z, int ldz, int C,
                       const float* Y0, int ldy0, const float* bias, int act, float* Y, int ldy, hipStream_t s) {
    constexpr size_t kLdsBudget = 64 * 1024;  // per workgroup: keeps >= 2 workgroups per CU
    const int zs = (C == 1) ? 1 : C + VEC;    // pad one vector: staggers rows across LDS banks
    const size_t zbytes = (size_t)((max_nodes * zs + 3) & ~3) * sizeof(float);
    static const int force_global = env_int("DGCN_SPMM_GLOBAL", 0);
    if (graph_ptr && B > 0 && max_nodes > 0 && zbytes + 2048 <= kLdsBudget && !force_global) {
        static const int rows_env = env_int("DGCN_SPMM_ROWS", 0);
        int rows_per_tile = rows_env > 0 ? rows_env : max_nodes;
        if (rows_env <= 0) {
            // enough workgroups to fill 256 CUs a few times over, but never below 64 rows a tile
            while ((long)B * ceil_div(max_nodes, rows_per_tile) < 1024 && rows_per_tile > 64)
                rows_per_tile = (rows_per_tile + 1) / 2;
        }
        const int tiles = ceil_div(max_nodes, rows_per_tile);
        int csr_cap = (int)((kLdsBudget - zbytes) / 8);
        static const int cap_env = env_int("DGCN_SPMM_CSRCAP", -1);
        if (cap_env >= 0) csr_cap = min(csr_cap, cap_env);
        csr_cap &= ~3;
        const size_t lds = zbytes + (size_t)csr_cap * 8;
        TimedLaunch t("spmm", s);
        hipLaunchKernelGGL((k_spmm_lds<VEC, LPR>), dim3((unsigned)tiles * (unsigned)B), dim3(256), lds, s, S->row_ptr,
                           S->col_idx, S->values, graph_ptr, tiles, rows_per_tile, Z, ldz, C, zs, csr_cap, Y0, ldy0,
                           bias, act, Y, ldy);
        return check_launch("k_spmm_lds");
    }
    constexpr int kGroups = 256 / LPR;
    int blocks = ceil_div(S->num_rows, kGroups);
    blocks = min(blocks, 256 * 16);
    TimedLaunch t("spmm", s);
    hipLaunchKernelGGL((k_spmm_global<VEC, LPR>), dim3(blocks), dim3(256), 0, s, S->row_ptr, S->col_idx, S->values,
                       S->num_rows, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy);
    return check_launch("k_spmm_global");
}

int spmm_dispatch(const DgcnCsr* S, const int32_t* graph_ptr, int B, int max_nodes, const float* Z, int ldz, int C,
                  const float* Y0, int ldy0, const float* bias, int act, float* Y, int ldy, hipStream_t s) {
    const bool vec_ok = (C % 4 == 0) && (ldz % 4 == 0) && (ldy % 4 == 0) && (!Y0 || ldy0 % 4 == 0) &&
                        ((uintptr_t)Z % 16 == 0) && ((uintptr_t)Y % 16 == 0) && (!Y0 || (uintptr_t)Y0 % 16 == 0) &&
                        (!bias || (uintptr_t)bias % 16 == 0);
#define DGCN_SPMM_CASE(V, L) return launch_spmm<V, L>(S, graph_ptr, B, max_nodes, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy, s)
    if (vec_ok) {
        if (C <= 4) DGCN_SPMM_CASE(4, 1);
        if (C <= 8) DGCN_SPMM_CASE(4, 2);
        if (C <= 16) DGCN_SPMM_CASE(4, 4);
        if (C <= 32) DGCN_SPMM_CASE(4, 8);
        if (C <= 64) DGCN_SPMM_CASE(4, 16);
        if (C <= 128) DGCN_SPMM_CASE(4, 32);
        if (C <= 256) DGCN_SPMM_CASE(4, 64);
    } else {
        if (C <= 1) DGCN_SPMM_CASE(1, 1);
        if (C <= 2) DGCN_SPMM_CASE(1, 2);
        if (C <= 4) DGCN_SPMM_CASE(1, 4);
        if (C <= 8) DGCN_SPMM_CASE(1, 8);
        if (C <= 16) DGCN_SPMM_CASE(1, 16);
        if (C <= 32) DGCN_SPMM_CASE(1, 32);
        if (C <= 64) DGCN_SPMM_CASE(1, 64);
    }
#undef DGCN_SPMM_CASE
    return fail(DGCN_ERR_UNSUPPORTED, "dgcn_spmm_batch: feature width C=%d not supported", C);
}

}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_spmm_batch(const DgcnCsr* S, const int32_t* graph_ptr, int32_t num_graphs, int32_t max_nodes,
                               const float* Z, int32_t ldz, int32_t C, const float* Y0, int32_t ldy0,
                               const float* bias, int32_t act, float* Y, int32_t ldy, void* stream) {
    if (!S || !Z || !Y || !S->row_ptr || (S->nnz > 0 && (!S->col_idx || !S->values)))
        return fail(DGCN_ERR_ARG, "dgcn_spmm_batch: null argument");
    if (C <= 0 || ldz < C || ldy < C || (Y0 && ldy0 < C)) return fail(DGCN_ERR_ARG, "dgcn_spmm_batch: bad strides");
    if (S->num_rows <= 0) return DGCN_OK;
    return spmm_dispatch(S, graph_ptr, num_graphs, max_nodes, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy,
                         (hipStream_t)stream);
}
