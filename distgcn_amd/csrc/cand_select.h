// The rollout's candidates: the first `beam` undecided vertices of a graph under (priority desc, index asc) - the stable argsort of
// -gcn_wts (mwis_gdpg_call.py:624-626) - selected by ONE workgroup that holds the graph's priorities in registers.  Shared by
// general.hip (k_res_cand: a launch of its own), big.hip and wide.hip (the end of a residual step's launch).
// Every wave selects ITS first `beam` (a vertex per lane: each lane counts the wave's vertices ahead of it, one v_readlane pass;
// several vertices per lane: `beam` rounds of: best of my vertices not taken yet, wave-wide argmax - inside a row of sixteen
// lanes by DPP moves, across the four rows by v_readlane -, its owner marks it taken); the waves' lists - sorted, and
// together they contain the graph's first `beam` - are ranked against each other: a candidate's rank is the sum over the lists of
// the entries ahead of it, a binary search per list, all threads on (candidate, list) pairs.
#pragma once
#include "common.h"

namespace dgcn {

constexpr int kCandMaxBeam = 64;
constexpr int kCandPer = 10;  // vertices per thread at most: 9 600 vertices on 1 024 threads
// LDS scratch of cand_select<BLOCK>: [waves * 64] doubles, 2 x [waves * 64] ints, [waves] ints
__host__ __device__ constexpr size_t cand_scratch_bytes(int block) { return (size_t)(block / 64) * kCandMaxBeam * 16 + (size_t)(block / 64) * 4 + 16; }

__device__ __forceinline__ bool cand_ahead(double p, int v, double q, int u) {  // is (p, v) ahead of (q, u)?  (both somebody)
    return p > q || (p == q && v < u);
}
template <int CTRL>
__device__ __forceinline__ void cand_dpp_max(double& p, int& v) {
    const int lo = __double2loint(p), hi = __double2hiint(p);
    const int olo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    const int ohi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    const int ov = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
    const double op = __hiloint2double(ohi, olo);
    if (ov >= 0 && (v < 0 || cand_ahead(op, ov, p, v))) { p = op; v = ov; }
}

// pv[i] / bit i of `have`: priority of vertex threadIdx.x + i * BLOCK and whether it takes part (undecided), i < per (uniform).
// cid: the graph's [64] candidate slots in global memory, all -1 on entry (the caller's business); scratch: cand_scratch_bytes(BLOCK)
// bytes of LDS, 8-byte aligned; cid_lds: null, or [64] slots in LDS that get the same list (all -1 on entry, outside `scratch`;
// readable behind the caller's next barrier).  Every thread of the workgroup must call (barriers inside).
template <int BLOCK>
__device__ __forceinline__ void cand_select(const double (&pv)[kCandPer], unsigned have, int per, int beam, int32_t* cid, unsigned char* scratch,
                                            int32_t* cid_lds = nullptr) {
    constexpr int kW = BLOCK / 64;
    double* wl_p = reinterpret_cast<double*>(scratch);
    int* wl_v = reinterpret_cast<int*>(wl_p + kW * kCandMaxBeam);
    int* rank = wl_v + kW * kCandMaxBeam;
    int* wl_n = rank + kW * kCandMaxBeam;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < kW * kCandMaxBeam; c += BLOCK) rank[c] = 0;
    int mine_n = 0;
    if (per == 1) {
        // A vertex per lane (graphs of up to BLOCK vertices): a lane's place in its wave's list is the number of the wave's
        // undecided vertices ahead of it - one pass over them with v_readlane, ~7 instructions each - instead of `beam` rounds
        // of a wave-wide argmax (~120 instructions each; sixteen waves on four SIMDs: 13 of a one-layer rollout step's 47 us).
        const bool on = (have & 1u) != 0u;
        const double p = pv[0];
        const unsigned long long onmask = __ballot(on);
        int ahead = 0;
        for (unsigned long long m = onmask; m; m &= m - 1ull) {  // (uniform)
            const int l = __ffsll((long long)m) - 1;
            const double op = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(p), l), __builtin_amdgcn_readlane(__double2loint(p), l));
            ahead += (op > p) || (op == p && l < lane);
        }
        if (on && ahead < beam) { wl_p[wave * kCandMaxBeam + ahead] = p; wl_v[wave * kCandMaxBeam + ahead] = (int)threadIdx.x; }
        mine_n = min(__popcll(onmask), beam);
    } else
    for (int it = 0; it < beam; ++it) {
        double bp = 0.0;
        int bv = -1;
#pragma unroll
        for (int i = 0; i < kCandPer; ++i)  // ascending vertex index: the first maximum stays
            if (i < per) {  // (uniform: a graph of up to BLOCK vertices has one vertex per thread)
                if (((have >> i) & 1u) && (bv < 0 || pv[i] > bp)) { bp = pv[i]; bv = (int)threadIdx.x + i * BLOCK; }
            }
        cand_dpp_max<0xB1>(bp, bv);   // quad_perm [1, 0, 3, 2]
        cand_dpp_max<0x4E>(bp, bv);   // quad_perm [2, 3, 0, 1]
        cand_dpp_max<0x141>(bp, bv);  // row_half_mirror
        cand_dpp_max<0x140>(bp, bv);  // row_mirror: every lane holds its row's best
        double wp = 0.0;
        int wv = -1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rv = __builtin_amdgcn_readlane(bv, 16 * r);
            const double rp = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bp), 16 * r), __builtin_amdgcn_readlane(__double2loint(bp), 16 * r));
            if (rv >= 0 && (wv < 0 || cand_ahead(rp, rv, wp, wv))) { wp = rp; wv = rv; }
        }
        if (wv < 0) break;  // (wave-uniform: this wave has no undecided vertex left)
        if ((wv % BLOCK) == (int)threadIdx.x) have &= ~(1u << (wv / BLOCK));
        if (lane == 0) { wl_p[wave * kCandMaxBeam + it] = wp; wl_v[wave * kCandMaxBeam + it] = wv; }
        mine_n = it + 1;
    }
    if (lane == 0) wl_n[wave] = mine_n;
    __syncthreads();
    for (int idx = threadIdx.x; idx < kW * beam * kW; idx += BLOCK) {
        const int w2 = idx % kW, c = idx / kW;
        const int w = c / beam, k = c - w * beam;
        if (k >= wl_n[w]) continue;
        const double p = wl_p[w * kCandMaxBeam + k];
        const int v = wl_v[w * kCandMaxBeam + k];
        int lo = 0, hi = wl_n[w2];  // entries [0, lo) of list w2 are ahead of the candidate, [hi, ..) are not
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (cand_ahead(wl_p[w2 * kCandMaxBeam + mid], wl_v[w2 * kCandMaxBeam + mid], p, v)) lo = mid + 1; else hi = mid;
        }
        if (lo) atomicAdd(&rank[w * kCandMaxBeam + k], lo);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < kW * beam; c += BLOCK) {
        const int w = c / beam, k = c - w * beam;
        if (k < wl_n[w] && rank[w * kCandMaxBeam + k] < beam) {
            cid[rank[w * kCandMaxBeam + k]] = wl_v[w * kCandMaxBeam + k];
            if (cid_lds) cid_lds[rank[w * kCandMaxBeam + k]] = wl_v[w * kCandMaxBeam + k];
        }
    }
}

}  // namespace dgcn
