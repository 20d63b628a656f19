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

// (col, val) of one nonzero packed in 8 bytes: the LDS copy is read with one ds_read_b64.
struct __attribute__((aligned(8))) ColVal { int col; float val; };

__device__ __forceinline__ float4 vshfl_xor(float4 a, int off) {
    return make_float4(__shfl_xor(a.x, off), __shfl_xor(a.y, off), __shfl_xor(a.z, off), __shfl_xor(a.w, off));
}
__device__ __forceinline__ float vshfl_xor(float a, int off) { return __shfl_xor(a, off); }

// Row sum with the row's nonzeros dealt round-robin to G lane groups (group g takes entries
// rs+g, rs+g+G, ...: a sequential fmaf chain each), then a butterfly over the groups:
//   for off = G/2 .. 1:  p[g] = p[g] + p[g ^ off]
// G = 1 is the plain sequential chain.  `meta(j)` returns the (offset-of-Z-row, value) of entry j.
// Every lane of the wave must call this (the butterfly is a wave-wide shuffle); lanes without work
// pass rs == re.
template <int VEC, int G, typename Meta, typename ZRow>
__device__ __forceinline__ typename VecT<VEC>::type row_sum_split(Meta meta, int rs, int re, int g, int foff,
                                                                  int lane_stride, ZRow zrow) {
    using V = typename VecT<VEC>::type;
    V acc = vzero<VEC>();
    int j = rs + g;
    for (; j + G < re; j += 2 * G) {
        const ColVal m0 = meta(j), m1 = meta(j + G);
        const V z0 = *reinterpret_cast<const V*>(zrow(m0.col) + foff);
        const V z1 = *reinterpret_cast<const V*>(zrow(m1.col) + foff);
        acc = vfma(m0.val, z0, acc);
        acc = vfma(m1.val, z1, acc);
    }
    if (j < re) {
        const ColVal m = meta(j);
        acc = vfma(m.val, *reinterpret_cast<const V*>(zrow(m.col) + foff), acc);
    }
#pragma unroll
    for (int off = G / 2; off >= 1; off >>= 1) acc = vadd(acc, vshfl_xor(acc, off * lane_stride));
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
// LDS variant: blockIdx.x = graph * tiles + tile.  BLOCK threads = BLOCK/LPR row groups.
// The three staging streams (Z slice, tile (col,val), nothing else) are issued as batches of
// independent loads (U per thread in flight) before their LDS stores: at ~1 us per dependent
// HBM/L2 round trip a one-load-per-iteration loop would dominate the kernel.
template <int U, int BLOCK, typename T, typename LoadF, typename StoreF>
__device__ __forceinline__ void staged_copy(int total, LoadF load, StoreF store) {
    for (int base = threadIdx.x; base < total; base += BLOCK * U) {
        T tmp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * BLOCK;
            if (i < total) tmp[u] = load(i);  // (clamping the index instead of predicating the load measured 5 % slower here)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = base + u * BLOCK;
            if (i < total) store(i, tmp[u]);
        }
    }
}

// F64: the row sum is ONE fma chain in double (G must be 1), out = (float)((double)y0 + sum [+ (double)bias]) - the
// contract of layer index 0 (k_spmm_f64acc below is the plain-kernel form of the same arithmetic).
template <int VEC, int G, typename Meta, typename ZRow>
__device__ __forceinline__ void row_sum_f64(Meta meta, int rs, int re, int foff, ZRow zrow, double (&acc)[VEC]) {
    using V = typename VecT<VEC>::type;
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.0;
    int j = rs;
    for (; j + 1 < re; j += 2) {  // two entries' loads in flight, chain order unchanged
        const ColVal m0 = meta(j), m1 = meta(j + 1);
        const V z0 = *reinterpret_cast<const V*>(zrow(m0.col) + foff);
        const V z1 = *reinterpret_cast<const V*>(zrow(m1.col) + foff);
        const float* f0 = reinterpret_cast<const float*>(&z0);
        const float* f1 = reinterpret_cast<const float*>(&z1);
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] = fma((double)m0.val, (double)f0[c], acc[c]);
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] = fma((double)m1.val, (double)f1[c], acc[c]);
    }
    if (j < re) {
        const ColVal m = meta(j);
        const V z = *reinterpret_cast<const V*>(zrow(m.col) + foff);
        const float* f = reinterpret_cast<const float*>(&z);
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] = fma((double)m.val, (double)f[c], acc[c]);
    }
}

template <int VEC, int LPR, int G, int BLOCK, bool F64 = false>
__global__ __launch_bounds__(BLOCK) void k_spmm_lds(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                    const float* __restrict__ values, const int32_t* __restrict__ graph_ptr,
                                                    int tiles, int rows_per_tile, const float* __restrict__ Z, int ldz,
                                                    int C, int zs, int csr_cap, const float* __restrict__ Y0, int ldy0,
                                                    const float* __restrict__ bias, int act, float* __restrict__ Y,
                                                    int ldy) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using V = typename VecT<VEC>::type;
    const int g = blockIdx.x / tiles;
    const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1];
    const int r0 = n0 + (blockIdx.x % tiles) * rows_per_tile;
    if (r0 >= n1) return;
    const int r1 = min(r0 + rows_per_tile, n1);
    const int ng = n1 - n0;
    float* zsm = smem;
    ColVal* cvs = reinterpret_cast<ColVal*>(smem + (size_t)((ng * zs + 3) & ~3));
    const int e0 = row_ptr[r0], e1 = row_ptr[r1];
    const bool csr_in_lds = (e1 - e0) <= csr_cap;
    // ---- stage this graph's slice of Z (coalesced VEC-wide loads, LDS row stride zs) ...
    {
        const int per_row = C / VEC;
        staged_copy<4, BLOCK, V>(
            ng * per_row,
            [&](int i) { const int row = i / per_row, q = i - row * per_row;
                         return *reinterpret_cast<const V*>(Z + (size_t)(n0 + row) * ldz + q * VEC); },
            [&](int i, V v) { const int row = i / per_row, q = i - row * per_row;
                              *reinterpret_cast<V*>(zsm + row * zs + q * VEC) = v; });
    }
    // ---- ... and the tile's (col, val) range, columns as graph-local ids
    if (csr_in_lds) {
        staged_copy<4, BLOCK, ColVal>(
            e1 - e0, [&](int i) { ColVal m; m.col = (col_idx[e0 + i] - n0) * zs; m.val = values[e0 + i]; return m; },
            [&](int i, ColVal m) { cvs[i] = m; });  // col pre-multiplied by the LDS row stride
    }
    // ---- row bounds and the "+ Y0" operand of this thread's rows: issued now, together with the staging
    // loads, so the row loop below starts without another dependent global round trip.
    // Lane map inside a wave: lane = g * (R * LPR) + r * LPR + sub  (R = 64 / (LPR * G) rows per wave).
    constexpr int R = 64 / (LPR * G);
    constexpr int kSlots = (BLOCK / 64) * R;  // rows in flight per pass
    constexpr int kPre = 4;                   // passes held in registers (tile <= kPre * kSlots rows, by launch)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rw = (lane / LPR) % R, gq = lane / (LPR * R);
    const int slot = wave * R + rw;
    const int foff = sub * VEC;
    const bool lane_on = foff < C;
    int rs[kPre], re[kPre];
    V y0[kPre];
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        const int v = r0 + slot + k * kSlots;
        rs[k] = re[k] = 0;
        y0[k] = vzero<VEC>();
        if (v < r1 && lane_on) {
            rs[k] = row_ptr[v];
            re[k] = row_ptr[v + 1];
            if (Y0 && gq == 0) y0[k] = *reinterpret_cast<const V*>(Y0 + (size_t)v * ldy0 + foff);
        }
    }
    V bv = vzero<VEC>();
    if (bias && lane_on) bv = *reinterpret_cast<const V*>(bias + foff);
    __syncthreads();
    auto zrow_scaled = [&](int off) -> const float* { return zsm + off; };
    auto meta_lds = [&](int j) -> ColVal { return cvs[j - e0]; };
    auto meta_glob = [&](int j) -> ColVal { ColVal m; m.col = (col_idx[j] - n0) * zs; m.val = values[j]; return m; };
#pragma unroll
    for (int k = 0; k < kPre; ++k) {
        if (r0 + wave * R + k * kSlots >= r1) break;  // wave-uniform: the whole wave is past the tile
        const int v = r0 + slot + k * kSlots;
        if constexpr (F64) {
            static_assert(G == 1, "the double chain is sequential");
            double accd[VEC];
            if (csr_in_lds) row_sum_f64<VEC, G>(meta_lds, rs[k], re[k], foff, zrow_scaled, accd);
            else row_sum_f64<VEC, G>(meta_glob, rs[k], re[k], foff, zrow_scaled, accd);
            if (v < r1 && lane_on) {
                V out;
                float* o = reinterpret_cast<float*>(&out);
                const float* yy = reinterpret_cast<const float*>(&y0[k]);
                const float* bb = reinterpret_cast<const float*>(&bv);
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    double d = accd[c];
                    if (Y0) d = (double)yy[c] + d;
                    if (bias) d = d + (double)bb[c];
                    o[c] = apply_act((float)d, act);
                }
                *reinterpret_cast<V*>(Y + (size_t)v * ldy + foff) = out;
            }
        } else {
        V acc;
        if (csr_in_lds) acc = row_sum_split<VEC, G>(meta_lds, rs[k], re[k], gq, foff, LPR * R, zrow_scaled);
        else acc = row_sum_split<VEC, G>(meta_glob, rs[k], re[k], gq, foff, LPR * R, zrow_scaled);
        if (v < r1 && lane_on && gq == 0) {
            if (Y0) acc = vadd(y0[k], acc);
            if (bias) acc = vadd(acc, bv);
            *reinterpret_cast<V*>(Y + (size_t)v * ldy + foff) = vact(acc, act);
        }
        }
    }
}

// (Tried in round 3 and dropped: a persistent variant - a fixed grid of two workgroups per CU walking graphs g, g + grid, ..,
// the next graph's Z slice / pairs / row bounds prefetched into registers while the current graph's rows gather from LDS,
// node and entry ranges fetched two and three graphs ahead, "+ Y0" operands requested first so that waiting for them does
// not mean waiting for the prefetch (vmcnt retires in order).  131 us against 105 - 109 us for the 4 000-graph launch:
// at 64 VGPRs (two 1024-thread workgroups per CU) the prefetch registers spill, and what the one-graph-per-workgroup
// kernel gets for free - 16 000 workgroups in every phase of their lives at once - the pipeline has to build by hand.)
// (Second attempt, later in round 3, with what the first one lacked: ONE 1024-thread workgroup per CU (128 VGPRs: nothing
// spilled), two LDS buffers, the next graph's Z slice by LDS-DMA (global_load_lds_dwordx4 as an asm statement - through the
// builtin the compiler orders the gather phase's first ds_read behind the DMA with s_waitcnt vmcnt(0) -; its pairs, row
// bounds and "+ Y0" operands through registers that nothing reads before the current graph's rows are done; waits said with
// __builtin_amdgcn_s_waitcnt so that the compiler's bookkeeping sees them), bounds of all the workgroup's graphs fetched
// once.  Bit-exact, and the gather phase has no vector-memory wait left in it - yet 107.3 us against 105.4 us on the
// 4 000-graph launch and 20.0 against 18.3 us on rotating 500-graph launches (tools/runs/r03_gpu31.sh).  A graph is 87 KB in
// and 26 KB out, two buffers are all the LDS holds (3 x 61 KB > 160 KB), so the pipeline is one graph deep: an iteration
// lasts what one graph's loads take from issue to arrival (~5 us at a CU's share of the HBM) plus the drain of its stores,
// not the ~3 us of its gathers - which is what two co-resident workgroups of the plain kernel already achieve between them.)
// ---------------------------------------------------------------------------------------------
// Global-gather variant: no graph structure needed.
template <int VEC, int LPR, int G>
__global__ __launch_bounds__(256) void k_spmm_global(const int32_t* __restrict__ row_ptr,
                                                     const int32_t* __restrict__ col_idx,
                                                     const float* __restrict__ values, int num_rows,
                                                     const float* __restrict__ Z, int ldz, int C,
                                                     const float* __restrict__ Y0, int ldy0,
                                                     const float* __restrict__ bias, int act,
                                                     float* __restrict__ Y, int ldy) {
    using V = typename VecT<VEC>::type;
    constexpr int R = 64 / (LPR * G);
    constexpr int kSlots = 4 * R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, rw = (lane / LPR) % R, gq = lane / (LPR * R);
    const int foff = sub * VEC;
    const bool lane_on = foff < C;
    auto zrow = [&](int u) -> const float* { return Z + (size_t)u * ldz; };
    auto meta = [&](int j) -> ColVal { ColVal m; m.col = col_idx[j]; m.val = values[j]; return m; };
    for (int base = blockIdx.x * kSlots; base < num_rows; base += gridDim.x * kSlots) {
        const int v = base + wave * R + rw;
        int rs = 0, re = 0;
        if (v < num_rows && lane_on) { rs = row_ptr[v]; re = row_ptr[v + 1]; }
        V acc = row_sum_split<VEC, G>(meta, rs, re, gq, foff, LPR * R, zrow);
        if (v < num_rows && lane_on && gq == 0) epilogue_store<VEC>(acc, v, foff, Y0, ldy0, bias, act, Y, ldy);
    }
}

// ---------------------------------------------------------------------------------------------
// Row sums carried in double (the contract of layer index 0, DGCN_PRECISE in include/dgcn.h): ONE fma chain over the row's
// entries in CSR order, out = (float)((double)y0 + acc [+ (double)bias]), activation in float32.  Runs once per forward
// of the layer-by-layer path: a plain global-gather kernel, LPR lanes x VEC features per row.
template <int VEC>
__global__ __launch_bounds__(256) void k_spmm_f64acc(const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                     const float* __restrict__ values, int num_rows, const float* __restrict__ Z,
                                                     int ldz, int C, const float* __restrict__ Y0, int ldy0,
                                                     const float* __restrict__ bias, int act, float* __restrict__ Y, int ldy) {
    const int per_row = (C + VEC - 1) / VEC;
    const long total = (long)num_rows * per_row;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int v = (int)(i / per_row), foff = (int)(i - (long)v * per_row) * VEC;
        double acc[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) acc[c] = 0.0;
        const int rs = row_ptr[v], re = row_ptr[v + 1];
        for (int j = rs; j < re; ++j) {
            const double a = (double)values[j];
            const float* z = Z + (size_t)col_idx[j] * ldz + foff;
            if constexpr (VEC == 4) {
                const float4 q = *reinterpret_cast<const float4*>(z);
                acc[0] = fma(a, (double)q.x, acc[0]); acc[1] = fma(a, (double)q.y, acc[1]);
                acc[2] = fma(a, (double)q.z, acc[2]); acc[3] = fma(a, (double)q.w, acc[3]);
            } else {
                acc[0] = fma(a, (double)z[0], acc[0]);
            }
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            double o = acc[c];
            if (Y0) o = (double)Y0[(size_t)v * ldy0 + foff + c] + o;
            if (bias) o = o + (double)bias[foff + c];
            Y[(size_t)v * ldy + foff + c] = apply_act((float)o, act);
        }
    }
}

static int spmm_f64acc_plain(const DgcnCsr* S, const float* Z, int ldz, int C, const float* Y0, int ldy0, const float* bias, int act,
                             float* Y, int ldy, hipStream_t s) {
    if (S->num_rows <= 0) return DGCN_OK;
    const bool vec = (C % 4 == 0) && (ldz % 4 == 0) && ((uintptr_t)Z % 16 == 0);
    const long total = (long)S->num_rows * (vec ? C / 4 : C);
    const int blocks = (int)min((total + 255) / 256, (long)256 * 32);
    TimedLaunch t("spmm", s);
    if (vec) DGCN_LAUNCH(t, (k_spmm_f64acc<4>), dim3(blocks), dim3(256), 0, s, S->row_ptr, S->col_idx, S->values, S->num_rows, Z, ldz, C,
                         Y0, ldy0, bias, act, Y, ldy);
    else DGCN_LAUNCH(t, (k_spmm_f64acc<1>), dim3(blocks), dim3(256), 0, s, S->row_ptr, S->col_idx, S->values, S->num_rows, Z, ldz, C,
                     Y0, ldy0, bias, act, Y, ldy);
    return check_launch("k_spmm_f64acc");
}




template <int VEC, int LPR, int G, int BLOCK, bool F64 = false>
static int launch_spmm_lds(const DgcnCsr* S, const int32_t* graph_ptr, int B, int tiles, int rows_per_tile,
                           const float* Z, int ldz, int C, int zs, int csr_cap, size_t lds, const float* Y0, int ldy0,
                           const float* bias, int act, float* Y, int ldy, hipStream_t s) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmm_lds<VEC, LPR, G, BLOCK, F64>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return fail(DGCN_ERR_LAUNCH, "k_spmm_lds: cannot reserve %zu bytes of LDS", lds);
    }
    TimedLaunch t("spmm", s);
    DGCN_LAUNCH(t, (k_spmm_lds<VEC, LPR, G, BLOCK, F64>), dim3((unsigned)tiles * (unsigned)B), dim3(BLOCK), lds, s,
                       S->row_ptr, S->col_idx, S->values, graph_ptr, tiles, rows_per_tile, Z, ldz, C, zs, csr_cap, Y0,
                       ldy0, bias, act, Y, ldy);
    return check_launch("k_spmm_lds");
}

template <int VEC, int LPR, int G>
static int launch_spmm(const DgcnCsr* S, const int32_t* graph_ptr, int B, int max_nodes, const float* Z, int ldz, int C,
                       const float* Y0, int ldy0, const float* bias, int act, float* Y, int ldy, hipStream_t s, bool f64 = false) {
    constexpr size_t kLdsMax = 150 * 1024;    // one workgroup may own (almost) the whole 160 KB LDS
    // LDS row stride of the staged Z slice = C + pad floats.  Measured on working sets beyond the Infinity Cache
    // (tools/tune_spmm_hbm.py, 4 000 ER graphs per launch): no padding 103.3 us, one vector 107.9 us, two 107.4 us -
    // the denser slice (25.6 instead of 28.8 KB per graph) is worth more than the staggered banks.
    const int pad = opt(OPT_SPMM_PAD);
    const int zs = (C == 1) ? 1 : C + (pad / VEC) * VEC;
    const size_t zbytes = (size_t)((max_nodes * zs + 3) & ~3) * sizeof(float);
    const int force_global = opt(OPT_SPMM_GLOBAL);
    if (graph_ptr && B > 0 && max_nodes > 0 && zbytes + 4096 <= kLdsMax && !force_global && (!f64 || (G == 1 && VEC == 4))) {
        // Tile = the whole graph unless the batch is too small to fill the chip; threads per
        // workgroup chosen so that every row group has about two rows.
        const int rows_env = opt(OPT_SPMM_ROWS);
        int rows_per_tile = rows_env > 0 ? rows_env : max_nodes;
        if (rows_env <= 0)
            while ((long)B * ceil_div(max_nodes, rows_per_tile) < 256 && rows_per_tile > 64)
                rows_per_tile = (rows_per_tile + 1) / 2;
        int block = opt(OPT_SPMM_BLOCK);
        if (block != 256 && block != 512 && block != 1024) {
            const int want_slots = (rows_per_tile + 1) / 2;  // about two passes per row slot
            block = want_slots * LPR * G <= 256 ? 256 : (want_slots * LPR * G <= 512 ? 512 : 1024);
        }
        // the kernel keeps at most 4 rows per row group in registers: shrink the tile to fit
        if (rows_per_tile > 4 * (block / (LPR * G))) rows_per_tile = 4 * (block / (LPR * G));
        const int tiles = ceil_div(max_nodes, rows_per_tile);
        // (col,val) slots: the exact per-graph maximum when known, else 1.15x the tile's average share of the nonzeros; a denser tile falls back to
        // reading its CSR range from global memory inside the kernel (still correct).
        const double avg_row = (double)S->nnz / (double)max(S->num_rows, 1);
        long cap = (long)(avg_row * rows_per_tile * 1.15) + 128;
        if (S->max_graph_nnz > 0 && tiles == 1) cap = S->max_graph_nnz;  // exact bound known
        const int cap_env = opt(OPT_SPMM_CSRCAP);
        if (cap_env >= 0) cap = cap_env;
        // prefer two workgroups per CU (<= 78 KB each) when the average tile still fits
        const long two_per_cu = ((long)78 * 1024 - (long)zbytes) / 8;
        if (cap > two_per_cu && two_per_cu >= (long)(avg_row * rows_per_tile * 1.15) + 128) cap = two_per_cu;
        cap = min(cap, (long)((kLdsMax - zbytes) / 8));
        const int csr_cap = (int)(cap & ~3L);
        const size_t lds = zbytes + (size_t)csr_cap * 8;
#define DGCN_SPMM_LDS(BL)                                                                                          \
    do {                                                                                                           \
        if constexpr (G == 1 && VEC == 4)                                                                          \
            if (f64) return launch_spmm_lds<VEC, LPR, 1, BL, true>(S, graph_ptr, B, tiles, rows_per_tile, Z, ldz, C, zs, csr_cap, lds, \
                                                                   Y0, ldy0, bias, act, Y, ldy, s);                \
        return launch_spmm_lds<VEC, LPR, G, BL>(S, graph_ptr, B, tiles, rows_per_tile, Z, ldz, C, zs, csr_cap, lds, Y0, ldy0, \
                                                bias, act, Y, ldy, s);                                             \
    } while (0)
        if (block == 256) DGCN_SPMM_LDS(256);
        if (block == 512) DGCN_SPMM_LDS(512);
        DGCN_SPMM_LDS(1024);
#undef DGCN_SPMM_LDS
    }
    if (f64) return spmm_f64acc_plain(S, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy, s);
    constexpr int kSlots = 256 / (LPR * G);
    int blocks = ceil_div(S->num_rows, kSlots);
    blocks = min(blocks, 256 * 16);
    TimedLaunch t("spmm", s);
    DGCN_LAUNCH(t, (k_spmm_global<VEC, LPR, G>), dim3(blocks), dim3(256), 0, s, S->row_ptr, S->col_idx, S->values,
                       S->num_rows, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy);
    return check_launch("k_spmm_global");
}

// The split factor G is part of the arithmetic (it fixes the summation order), so it depends on
// the feature width only - never on the data.  DGCN_SPMM_SPLIT overrides it for tuning.
int spmm_split_for(int C) {
    const int env = opt(OPT_SPMM_SPLIT);
    int g = env > 0 ? env : 1;  // measured on MI355X (C=32, ER and BA batches): the plain chain is fastest
    const bool vec = (C % 4 == 0);
    int lpr = 1;
    while (lpr * (vec ? 4 : 1) < C) lpr *= 2;
    while (g > 1 && lpr * g > 64) g /= 2;
    if (g != 1 && g != 2 && g != 4 && g != 8) g = 1;
    return g;
}

int spmm_dispatch(const DgcnCsr* S, const int32_t* graph_ptr, int B, int max_nodes, const float* Z, int ldz, int C,
                  const float* Y0, int ldy0, const float* bias, int act, float* Y, int ldy, hipStream_t s) {
    const bool aligned = (ldz % 4 == 0) && (ldy % 4 == 0) && (!Y0 || ldy0 % 4 == 0) && ((uintptr_t)Z % 16 == 0) &&
                         ((uintptr_t)Y % 16 == 0) && (!Y0 || (uintptr_t)Y0 % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0);
    const bool vec = (C % 4 == 0) && aligned;
    const int G = spmm_split_for(C);
#define DGCN_SPMM_CASE(V, L)                                                                                          \
    do {                                                                                                              \
        if (G == 1) return launch_spmm<V, L, 1>(S, graph_ptr, B, max_nodes, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy, s); \
        if constexpr (L * 2 <= 64)                                                                                    \
            if (G == 2) return launch_spmm<V, L, 2>(S, graph_ptr, B, max_nodes, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy, s); \
        if constexpr (L * 4 <= 64)                                                                                    \
            if (G == 4) return launch_spmm<V, L, 4>(S, graph_ptr, B, max_nodes, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy, s); \
        if constexpr (L * 8 <= 64)                                                                                    \
            if (G == 8) return launch_spmm<V, L, 8>(S, graph_ptr, B, max_nodes, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy, s); \
        return fail(DGCN_ERR_ARG, "dgcn_spmm_batch: split %d does not fit width %d", G, C);                         \
    } while (0)
    if (vec) {
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

// layer index 0's aggregation: the LDS-staged kernel with the double chain when the width allows 16-byte lanes (and the
// graphs fit), the plain kernel otherwise
int spmm_f64acc_dispatch(const DgcnCsr* S, const int32_t* graph_ptr, int B, int max_nodes, const float* Z, int ldz, int C,
                         const float* Y0, int ldy0, const float* bias, int act, float* Y, int ldy, hipStream_t s) {
    const bool aligned = (ldz % 4 == 0) && (ldy % 4 == 0) && (!Y0 || ldy0 % 4 == 0) && ((uintptr_t)Z % 16 == 0) &&
                         ((uintptr_t)Y % 16 == 0) && (!Y0 || (uintptr_t)Y0 % 16 == 0) && (!bias || (uintptr_t)bias % 16 == 0);
    if ((C % 4 == 0) && aligned && graph_ptr && B > 0) {
#define DGCN_SPMM_F64(L) return launch_spmm<4, L, 1>(S, graph_ptr, B, max_nodes, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy, s, true)
        if (C <= 4) DGCN_SPMM_F64(1);
        if (C <= 8) DGCN_SPMM_F64(2);
        if (C <= 16) DGCN_SPMM_F64(4);
        if (C <= 32) DGCN_SPMM_F64(8);
        if (C <= 64) DGCN_SPMM_F64(16);
        if (C <= 128) DGCN_SPMM_F64(32);
#undef DGCN_SPMM_F64
    }
    return spmm_f64acc_plain(S, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy, s);
}

}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_spmm_split(int32_t C) { return C > 0 ? spmm_split_for(C) : 1; }

extern "C" int dgcn_spmm_f64acc_batch(const DgcnCsr* S, const int32_t* graph_ptr, int32_t num_graphs, int32_t max_nodes,
                                      const float* Z, int32_t ldz, int32_t C, const float* Y0, int32_t ldy0,
                                      const float* bias, int32_t act, float* Y, int32_t ldy, void* stream) {
    if (!S || !Z || !Y || !S->row_ptr || (S->nnz > 0 && (!S->col_idx || !S->values)))
        return fail(DGCN_ERR_ARG, "dgcn_spmm_f64acc_batch: null argument");
    if (C <= 0 || ldz < C || ldy < C || (Y0 && ldy0 < C)) return fail(DGCN_ERR_ARG, "dgcn_spmm_f64acc_batch: bad strides");
    if (S->num_rows <= 0) return DGCN_OK;
    return spmm_f64acc_dispatch(S, graph_ptr, num_graphs, max_nodes, Z, ldz, C, Y0, ldy0, bias, act, Y, ldy, (hipStream_t)stream);
}

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
