// Shared by big.hip (k_big: graphs up to 976 vertices, Z1 whole in LDS) and big2.hip (k_big2: up to 1 920 vertices, Z1 a feature
// half at a time): the launch arguments and the small device helpers.  See big.hip for the design.
#pragma once
#include "common.h"

namespace dgcn {

constexpr int kBigBlock = 1024;     // one graph per CU: graphs above 512 vertices; 512 threads (two graphs per CU) below
constexpr int kBigMaxNodes = 976;   // Z1 (128 B per vertex) + a 2 KB staging tile per wave + the row tables in 160 KB; 61 tiles
constexpr int kBigTilesPerWave = 4;  // at most: 64 tiles over 16 waves (k_big<BLOCK, TILES>: 2 where half of that is enough)
constexpr int kBigMaxLayers = 64;
constexpr int kBH = 32;
constexpr int kBigBins = 1024;    // one bin per entry count of a row in the row-order counting sort (4 KB of the staging tiles' space)

using bf32x4 = __attribute__((ext_vector_type(4))) float;
using bf64x4 = __attribute__((ext_vector_type(4))) double;

struct BigLayer {
    const float* bias;   // of the aggregation this entry stands for, or null
    const float* Wnext;  // [32][64] weights of the NEXT layer's transform, or null after the last hidden aggregation
    int32_t act, pad;
};

struct BigFront {          // layer index 0 on constant features + the transform of layer index 1 (both with chains in double)
    const float* W0;       // [cin][64]
    const float* bias0;
    const float* W1;       // [32][64]
    float x_const;
    int32_t cin, act0, pad;
};

struct BigArgs {
    const int32_t* graph_ptr;
    const int32_t* lrow;   // support L: row pointers, diagonal first (null with `arow`)
    const int32_t* lcol;   // global column ids
    const float* lval;
    // or the adjacency itself: L = I - D^-1/2 A D^-1/2 is formed while the records are written (gcn/utils.py:120-127, 258-274;
    // supports.hip's expression: (float)(-(dinv[deg u] * dinv[deg v])), diagonal 1.0f first) - no k_supports launch, no L in HBM
    const int32_t* arow;
    const int32_t* acol;
    const double* dinv;    // float64 d^-1/2 by degree
    int32_t table_len;
    // the local greedy search at the end of the launch (heuristics.py:77-116; lgs_rounds.h) - with `arow` only
    int32_t do_lgs, predict_mwis, lgs_cols_lds;
    const double* weights;
    uint8_t* state;
    int32_t* rounds;
    double* totals;
    // a residual step of dgcn_solve_residual_batch in the same launch (with `arow`, constant input features): `state` is the
    // running state (0 = undecided) - the residual graph's support is formed from the adjacency and the state while the records are
    // written, the greedy step runs on the undecided vertices (general.hip's k_res_count / _scan / _fill / _scatter, k_lgs and
    // k_res_central in one launch)
    int32_t residual, greedy_mode, max_rounds;  // greedy_mode 0 rounds (max_rounds), 1 central pick, 2 priorities only (rollout)
    int32_t* progress;
    int32_t* active;       // greedy_mode 2: [num_graphs] out
    double* prio_out;      // greedy_mode 2: [num_nodes] out
    int32_t* cid;          // greedy_mode 2: [num_graphs][64] the rollout's candidates (cand_select.h), or null (k_res_cand follows)
    int32_t beam;
    int32_t ahead_rounds;  // whole searches of k_big2: the rounds on ahead lists (lgs_rounds_ahead); 0: lgs_rounds (option "wide_ahead" = 0)
    int32_t roll_off;      // greedy_mode 2: byte offset of the LDS the completions and the pick run in (rollout_bits.h: the whole
    int32_t by_priority;   // step in this launch); 0: general.hip's k_lgs / k_res_pick launches follow.  by_priority: their order
    unsigned long long* tail_word;
    unsigned long long tail_tag;
    const float* Zin;      // front == 0: [num_nodes][64] Z0 | Z1 of layer index 1 (explicit input features: the caller ran
                           // layer 0 and the transform of layer 1 with the layer-by-layer kernels)
    BigFront first;        // front == 1
    const float* Wlast;    // [32][2] the last layer's weights (w0 | w1)
    const float* bias_last;
    float* scores;         // [num_nodes]
    int32_t front, act_last;
    uint2* rec;            // [num_graphs][rec_cap]
    float4* stash;         // k_big2 beyond twelve tiles per wave: [num_graphs][max_nodes * 4] lo sums between the two walks
    int32_t* status;
    int32_t num_graphs_diag;     // DGCN_DIAG builds only: graphs of the launch (the per-wave table sits behind the per-graph one)
    unsigned long long* stamps;  // DGCN_DIAG builds only (tools/stamp_big.py): [num_graphs][16] phase clocks of wave 0 (s_memtime), kept in registers and written once
    int32_t rec_cap, max_nodes, num_hidden;
    int32_t lds_cnt_off, lds_perm_off, lds_stage_off, lds_tab_off;  // byte offsets inside the dynamic LDS; the zero row sits at max_nodes * 128
    BigLayer layers[kBigMaxLayers];
};

// where the greedy search's LDS arrays start: behind the last layer's z1 array (float per vertex + the neutral slot)
__host__ __device__ __forceinline__ unsigned big_lgs_base(int max_nodes) { return (4u * (unsigned)(max_nodes + 1) + 15u) & ~15u; }

__device__ __forceinline__ int big_key(int row) { return (row >> 1) & 3; }
__device__ __forceinline__ unsigned big_word(int u) { return ((unsigned)u << 7) | ((unsigned)big_key(u) << 4); }

typedef __attribute__((address_space(3))) const bf32x4 big_lds_cf4;
__device__ __forceinline__ float4 big_lds_chunk(unsigned addr) {  // Z1 chunk at ABSOLUTE LDS byte address (bufB at LDS offset 0)
    const bf32x4 z = *reinterpret_cast<big_lds_cf4*>(addr);
    return make_float4(z[0], z[1], z[2], z[3]);
}
__device__ __forceinline__ float4 big_fma4(float a, float4 z, float4 acc) {
    acc.x = fmaf(a, z.x, acc.x); acc.y = fmaf(a, z.y, acc.y); acc.z = fmaf(a, z.z, acc.z); acc.w = fmaf(a, z.w, acc.w);
    return acc;
}
__device__ __forceinline__ float big_act(float x, int act) { return apply_act(x, act); }

struct BigRec4 { uint2 r0, r1, r2, r3; };  // the records of four consecutive trips of this lane
__device__ __forceinline__ void big_load_group(BigRec4& G, const char* p) {
    G.r0 = *reinterpret_cast<const uint2*>(p);
    G.r1 = *reinterpret_cast<const uint2*>(p + 512);
    G.r2 = *reinterpret_cast<const uint2*>(p + 1024);
    G.r3 = *reinterpret_cast<const uint2*>(p + 1536);
}

// `f64map`: fragments for the f64 MFMA, which returns rows 4 * reg + (lane >> 4) where the f32 one returns 4 * (lane >> 4) + reg:
// lane r feeds column 4 * (r & 3) + (r >> 2) of the tile and the accumulator again holds four CONSECUTIVE features per lane
__device__ __forceinline__ void big_load_bfrag(const float* W, float (&b)[8][4], bool f64map) {
    const int lane = threadIdx.x & 63;
    const int r0 = lane & 15, kq = lane >> 4;
    const int r = f64map ? 4 * (r0 & 3) + (r0 >> 2) : r0;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) b[s][ct] = W[(4 * s + kq) * 64 + ct * 16 + r];
}

}  // namespace dgcn
