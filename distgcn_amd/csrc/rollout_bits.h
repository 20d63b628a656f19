// The rollout step's completions and pick inside the step's own launch (solve_mwis_rollout, mwis_gdpg_call.py:629-659): for each
// of the first `beam` candidates c the residual graph minus c's closed neighbourhood is searched greedily (by weight, or by
// priority) and the candidate with the largest weight + completion total joins.  general.hip runs this as k_lgs on
// beam x num_graphs masked instances plus k_res_pick; here ONE workgroup - the one that just computed the graph's scores and
// candidates (k_wide1, k_big, k_big2) - runs all instances of its graph at once, an instance per BIT:
//   S[v] = live mask (bits 0..15: v is still undecided in instance i) | joined mask << 16
//   a vertex looks at its neighbours AHEAD of it in the order (key desc, index asc; the lists are compacted to the front of
//   each row of the graph's 16-bit column ids in LDS, once): an ahead-neighbour that joined kills it, none left alive lets it
//   join - for all sixteen instances with a handful of bit operations per neighbour.
// No rounds and no barriers while the instances run: every wave goes over its vertices until they are decided, and decisions of
// other waves are seen as they happen.  The result does not depend on the schedule: the set a greedy search by a total order
// returns is the unique independent set in which every excluded vertex has a member neighbour ahead of it (heuristics.py:13-35
// sweeps sequentially, :77-116 in synchronous rounds - same set; tests hold this path against k_lgs's rounds and the oracle).
// Totals: an instance per wave, lane-strided partial sums, then DPP moves inside the rows and v_readlane across them - a fixed order, equal run to run; they differ
// from k_lgs's tree by rounding only and the pick compares them with the reference's 1e-12 relative tolerance.
#pragma once
#include "common.h"
#include "wave_reduce.h"

namespace dgcn {

constexpr int kRollBeam = 16;  // instances per state word: rollouts with more candidates take general.hip's launches

// LDS the caller sets aside: S u32[max_nodes] | ahead counts u16[max_nodes] | start masks u16[max_nodes] | totals f64[16]
__host__ __device__ constexpr size_t roll_lds_bytes(int max_nodes) {
    return (((size_t)max_nodes * 4 + 15) & ~(size_t)15) + 2 * (((size_t)max_nodes * 2 + 15) & ~(size_t)15) + kRollBeam * 8;
}

struct RollArgs {
    int ng, n0, e0;           // vertices of the graph, its first vertex / entry in the batch
    const double* key;        // LDS [ng]: the completions' order key of every undecided vertex (weight, or priority)
    const uint8_t* st;        // LDS [ng]: the running state, 0 = undecided
    const int* rol;           // LDS [ng + 1]: row bounds as entry numbers of the batch
    uint16_t* cl;             // LDS: the graph's local column ids (>= ng: not a vertex), entry j at cl[j - e0]; REWRITTEN (ahead lists at the row fronts)
    const int32_t* cidl;      // LDS [>= beam]: the candidates, the list ends at the first negative entry
    int beam;                 // <= kRollBeam
    unsigned char* extra;     // LDS, roll_lds_bytes(max_nodes), 16-byte aligned
    int max_nodes;
    const double* wl;         // LDS [ng] or null: the vertices' weights when the caller has them there (key, when the order is by weight)
    const double* weights;    // global, never null (the entry point refuses a rollout without weights)
    uint8_t* state;           // global, in / out
    int32_t* rounds;          // global [num_graphs] or null
    double* totals;           // global [num_graphs] or null
};

// Every thread of the workgroup calls; barriers inside; the caller passes a barrier after everything in RollArgs is written.
template <int BLOCK>
__device__ __forceinline__ void rollout_bits(const RollArgs& r, int g) {
    constexpr int kW = BLOCK / 64;
    unsigned* S = reinterpret_cast<unsigned*>(r.extra);
    uint16_t* acnt = reinterpret_cast<uint16_t*>(r.extra + (((size_t)r.max_nodes * 4 + 15) & ~(size_t)15));
    double* tot = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(acnt) + 2 * (((size_t)r.max_nodes * 2 + 15) & ~(size_t)15));
    volatile unsigned* Sv = S;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int ng = r.ng, e0 = r.e0;
    int nc = 0;
    while (nc < r.beam && nc < kRollBeam && r.cidl[nc] >= 0) ++nc;  // (uniform)
    if (nc == 0) {  // (a graph that is active has an undecided vertex, hence a candidate; kept for safety)
        if (threadIdx.x == 0) {
            if (r.rounds) r.rounds[g] = 0;
            if (r.totals) r.totals[g] = 0.0;
        }
        return;
    }
    const unsigned full = (1u << nc) - 1u;
    uint16_t* m0 = acnt + (((size_t)r.max_nodes * 2 + 15) & ~(size_t)15) / 2;  // [ng] who is alive in which instance when the instances start
    // (no global memory from here to the state bytes at the end: a candidate's neighbourhood is read from the column lists in LDS
    // while they are still whole, the picked candidate's is recovered from the start masks)
    // ---- every undecided vertex: alive in every instance
    for (int v = threadIdx.x; v < ng; v += BLOCK) S[v] = r.st[v] == 0 ? full : 0u;
    __syncthreads();
    // ---- instance i: candidate i and its neighbours do not take part
    for (int i = wave; i < nc; i += kW) {
        const int c = r.cidl[i];
        for (int j = r.rol[c] - e0 + lane; j < r.rol[c + 1] - e0; j += 64) {
            const int u = r.cl[j];
            if (u < ng) atomicAnd(&S[u], ~(1u << i));
        }
        if (lane == 0) atomicAnd(&S[c], ~(1u << i));
    }
    __syncthreads();
    // ---- the start masks; every undecided vertex's neighbours ahead of it, compacted to the front of its row
    for (int v = threadIdx.x; v < ng; v += BLOCK) {
        int cnt = 0;
        if (r.st[v] == 0) {
            const double kv = r.key[v];
            const int rs = r.rol[v] - e0, re = r.rol[v + 1] - e0;
            for (int j = rs; j < re; ++j) {
                const int u = r.cl[j];
                if (u < ng && u != v && r.st[u] == 0) {
                    const double ku = r.key[u];
                    if (ku > kv || (ku == kv && u < v)) r.cl[rs + cnt++] = (uint16_t)u;  // (behind the read position: same thread, in order)
                }
            }
        }
        acnt[v] = (uint16_t)cnt;
        m0[v] = (uint16_t)S[v];
    }
    __syncthreads();
    // ---- the instances: a vertex dies where an ahead-neighbour has joined, joins where none of them is alive any more
    bool more;
    do {
        more = false;
        for (int v = threadIdx.x; v < ng; v += BLOCK) {
            unsigned mine = Sv[v];
            const unsigned live = mine & 0xffffu;
            if (!live) continue;
            unsigned seen = 0u, killed = 0u;
            const int rs = r.rol[v] - e0, n = acnt[v];
            int k = 0;
            for (; k + 3 < n; k += 4) {  // four column -> state chains in flight
                const int u0 = r.cl[rs + k], u1 = r.cl[rs + k + 1], u2 = r.cl[rs + k + 2], u3 = r.cl[rs + k + 3];
                const unsigned s0 = Sv[u0], s1 = Sv[u1], s2 = Sv[u2], s3 = Sv[u3];
                seen |= s0 | s1 | s2 | s3;
            }
            for (; k < n; ++k) seen |= Sv[r.cl[rs + k]];
            killed = seen >> 16;
            const unsigned die = live & killed, win = live & ~killed & ~(seen & 0xffffu);
            if (die | win) {
                mine = (live & ~die & ~win) | (((mine >> 16) | win) << 16);
                Sv[v] = mine;
            }
            more |= (mine & 0xffffu) != 0u;
        }
    } while (__any(more));
    __syncthreads();
    // ---- totals: instance i on wave i, a fixed order
    for (int i = wave; i < nc; i += kW) {
        double part = 0.0;
        if (r.wl) {
            for (int v = lane; v < ng; v += 64)
                if ((S[v] >> (16 + i)) & 1u) part += r.wl[v];
        } else {
            // (every weight asked for, joined or not: a load inside the branch is a global round trip per pass of the loop)
            for (int v = lane; v < ng; v += 64) {
                const double w = r.weights[r.n0 + v];
                part += ((S[v] >> (16 + i)) & 1u) ? w : 0.0;
            }
        }
        part = wave_sum_f64(part);  // (DPP inside the rows, v_readlane across them: a fixed order)
        if (lane == 0) tot[i] = (r.wl ? r.wl[r.cidl[i]] : r.weights[r.n0 + r.cidl[i]]) + part;
    }
    __syncthreads();
    // ---- the pick: the largest total; totals within 1e-12 relative count as tied and the first candidate wins
    // (np.isclose(cand, cand.max(), rtol=1e-12, atol=0), as in fused.hip and k_res_pick)
    double cand = lane < nc ? tot[lane] : -1.0 / 0.0;
    const double mx = wave_max_f64(cand);
    // (cand == mx: equal infinities are close for np.isclose; nobody close - NaN totals -: the first candidate, as in fused.hip /
    // tail.hip / k_res_pick: a step that picks nobody would leave the graph active for ever.  Every wave computes the same pick.)
    const unsigned long long tied = __ballot(lane < nc && (cand == mx || fabs(cand - mx) <= 1e-12 * fabs(mx)));
    const int best = nc > 0 ? (tied ? __ffsll((long long)tied) - 1 : 0) : -1;
    if (best < 0) {
        if (threadIdx.x == 0) {
            if (r.rounds) r.rounds[g] = 0;
            if (r.totals) r.totals[g] = 0.0;
        }
        return;
    }
    // the picked candidate joins, its undecided neighbours leave: exactly the vertices instance `best` started without
    const int c = r.cidl[best];
    for (int v = threadIdx.x; v < ng; v += BLOCK)
        if (r.st[v] == 0 && !((m0[v] >> best) & 1u)) r.state[r.n0 + v] = v == c ? 1 : 2;
    if (threadIdx.x == 0) {
        if (r.rounds) r.rounds[g] = 1;
        if (r.totals) r.totals[g] = r.wl ? r.wl[c] : r.weights[r.n0 + c];
    }
}

}  // namespace dgcn
