// The local greedy search's device-side core, shared by lgs.hip (k_lgs) and big.hip (k_big runs the search at the end of its
// launch): heuristics.py:77-116, synchronous rounds under the order (priority desc, index asc).  See lgs.hip for the mapping.
#pragma once
#include "common.h"

namespace dgcn {

struct LgsArgs {
    const int32_t* graph_ptr;
    const int32_t* row_ptr;
    const int32_t* col_idx;
    const double* prio;
    const float* scores;
    const double* weights;
    int max_rounds;
    uint8_t* state;
    int32_t* rounds;
    int64_t* stats;
    int32_t* overhead;
    const double* sum_weights;
    double* totals;
    int32_t* status;
    int max_nodes;   // LDS carve-up
    int cols_cap;    // 16-bit column slots in LDS (0 = read col_idx from global memory)
    // masked / multi-instance form (dgcn_lgs_masked_batch): instance k works on its own
    // [num_nodes] slice of state/init_state/overhead (and prio when prio_stride != 0) and its own
    // [num_graphs] slice of rounds/totals/stats, all over the SAME block-diagonal adjacency.
    const uint8_t* init_state;  // non-zero = vertex does not take part (kept as given in the output)
    int num_graphs, num_nodes;
    long prio_stride;
    const int32_t* active;      // [num_graphs] or null: 0 = leave this graph alone (rounds 0, total 0, state untouched) -
                                // the "nothing left / no positive weight left" graphs of a residual step (general.hip)
    // the rollout's instances (general.hip): init_state is the RUNNING state, shared by every instance ([num_nodes], no slice
    // per instance), and instance k of graph g starts from it minus the closed neighbourhood of vertex cand[g * 64 + k]
    // (mwis_gdpg_call.py:629-643; a negative entry: no such candidate, the instance has nothing to search) - the mask is made in
    // LDS, no launch writes it out first
    const int32_t* cand;
    int ahead_rounds;           // whole searches without statistics: lgs_rounds_ahead below (0: lgs_rounds; option "wide_ahead" = 0)
};

template <bool COLS_LDS>
__device__ __forceinline__ int nbr_at(const uint16_t* cl, const int32_t* cg, int j, int e0, int n0) {
    if (COLS_LDS) return cl[j - e0];
    return cg[j] - n0;
}

// One residual neighbour test: liveness byte, then the float64 priority compare of the total order
// (priority desc, index asc).  (A variant on precomputed 16-bit ranks, as in the fused kernel, was
// measured slower here: the O(N^2) ranking costs more than the cheaper tests save.)
__device__ __forceinline__ void nbr_test(int u, int v, double pv, const uint8_t* st, const double* pr, bool& lost,
                                         int& resid) {
    if (st[u] == 0) {
        const double pu = pr[u];
        lost |= (pu > pv) || (pu == pv && u < v);
        ++resid;
    }
}

// FLAG_OR: the "anybody still in?" vote through acc64[3] instead of __syncthreads_count - that one (ockl's workgroup reduction)
// brings a static LDS allocation with it, and k_big addresses its dynamic LDS from offset 0
template <int LPV, bool STATS, bool COLS_LDS, int BLOCK, bool FLAG_OR = false>
__device__ __forceinline__ void lgs_rounds(const LgsArgs& a, int g, int n0, int ng, int e0, double* pr, uint8_t* st,
                                           uint8_t* nw, const uint16_t* cl, unsigned long long* acc64, const int* ro) {
    constexpr int kVerts = BLOCK / LPV;
    const int lane = threadIdx.x & 63;
    const int slot = threadIdx.x / LPV, sub = threadIdx.x % LPV;
    const int gshift = lane & ~(LPV - 1);
    const unsigned long long gmask = (LPV == 64) ? ~0ull : ((1ull << LPV) - 1ull);
    const int passes = (ng + kVerts - 1) / kVerts;
    unsigned long long p2p = 0, bst = 0;
    int rounds = 0;
    int remaining = ng;
    if (a.init_state) {  // masked start: count the vertices that actually take part
        if (threadIdx.x == 0) acc64[2] = 0;
        __syncthreads();
        int c = 0;
        for (int v = threadIdx.x; v < ng; v += BLOCK) c += st[v] == 0;
        if (c) atomicAdd(&acc64[2], (unsigned long long)c);
        __syncthreads();
        remaining = (int)acc64[2];
        __syncthreads();
    }
    while (remaining > 0 && (a.max_rounds <= 0 || rounds < a.max_rounds)) {
        if (STATS && threadIdx.x == 0) bst += (unsigned long long)remaining;
        // ---------------- phase A: who wins this round
        for (int p = 0; p < passes; ++p) {
            const int v = p * kVerts + slot;
            const bool live = v < ng && st[v] == 0;
            bool lost = false;
            int resid = 0;
            if (live) {
                const double pv = pr[v];
                const int rs = ro[v], re = ro[v + 1];
                int j = rs + sub;
                for (; j + 3 * LPV < re; j += 4 * LPV) {  // four independent neighbour chains in flight
                    const int u0 = nbr_at<COLS_LDS>(cl, a.col_idx, j, e0, n0);
                    const int u1 = nbr_at<COLS_LDS>(cl, a.col_idx, j + LPV, e0, n0);
                    const int u2 = nbr_at<COLS_LDS>(cl, a.col_idx, j + 2 * LPV, e0, n0);
                    const int u3 = nbr_at<COLS_LDS>(cl, a.col_idx, j + 3 * LPV, e0, n0);
                    nbr_test(u0, v, pv, st, pr, lost, resid);
                    nbr_test(u1, v, pv, st, pr, lost, resid);
                    nbr_test(u2, v, pv, st, pr, lost, resid);
                    nbr_test(u3, v, pv, st, pr, lost, resid);
                }
                for (; j < re; j += LPV)
                    nbr_test(nbr_at<COLS_LDS>(cl, a.col_idx, j, e0, n0), v, pv, st, pr, lost, resid);
            }
            if (LPV > 1) {
                const unsigned long long m = __ballot(lost);
                lost = ((m >> gshift) & gmask) != 0ull;
            }
            if (STATS) {
#pragma unroll
                for (int off = 1; off < LPV; off <<= 1) resid += __shfl_xor(resid, off);
            }
            if (live && sub == 0) {
                nw[v] = lost ? 0 : 1;
                if (STATS) {
                    p2p += (unsigned long long)resid;
                    if (a.overhead) a.overhead[n0 + v] += resid + ((!lost && resid > 0) ? 1 : 0);
                }
            }
        }
        __syncthreads();
        if (FLAG_OR && threadIdx.x == 0) acc64[3] = 0;  // (every thread has read last round's vote: it decided to enter this round)
        // ---------------- phase B: winners join, their remaining neighbours are excluded
        for (int p = 0; p < passes; ++p) {
            const int v = p * kVerts + slot;
            if (v < ng && st[v] == 0 && nw[v]) {
                const int rs = ro[v], re = ro[v + 1];
                for (int j = rs + sub; j < re; j += LPV) {
                    const int u = nbr_at<COLS_LDS>(cl, a.col_idx, j, e0, n0);
                    if (st[u] == 0) st[u] = 2;  // same value from every writer: benign
                }
            }
        }
        __syncthreads();
        int mine = 0;
        for (int v = threadIdx.x; v < ng; v += BLOCK) {
            if (st[v] == 0) {
                if (nw[v]) st[v] = 1; else ++mine;
            }
            nw[v] = 0;
        }
        if constexpr (FLAG_OR) {
            if (mine > 0) acc64[3] = 1;  // (same value from every writer)
            __syncthreads();
            remaining = acc64[3] != 0 ? 1 : 0;
        } else {
            remaining = __syncthreads_count(mine > 0) ? 1 : 0;
        }
        if (remaining) {
            // exact count only matters for bst (stats); otherwise "some remain" is enough
            if (STATS) {
                if (threadIdx.x == 0) acc64[2] = 0;
                __syncthreads();
                if (mine) atomicAdd(&acc64[2], (unsigned long long)mine);
                __syncthreads();
                remaining = (int)acc64[2];
            }
        }
        ++rounds;
    }
    if (threadIdx.x == 0 && a.rounds) a.rounds[g] = rounds;
    if (STATS) {
        if (threadIdx.x == 0) { acc64[0] = 0; acc64[1] = 0; }
        __syncthreads();
        if (p2p) atomicAdd(&acc64[0], p2p);
        int members = 0;
        for (int v = threadIdx.x; v < ng; v += BLOCK) members += st[v] == 1;
        if (members) atomicAdd(&acc64[1], (unsigned long long)members);
        __syncthreads();
        if (threadIdx.x == 0 && a.stats) {
            a.stats[2 * g + 0] = (int64_t)acc64[0];
            a.stats[2 * g + 1] = (int64_t)(bst + acc64[1]);  // bst += len(mwis) (heuristics.py:208)
        }
    }
}

// ---- the same search on bit masks (graphs with a thread per vertex: ng <= BLOCK <= 1024) --------------------------------
// A live vertex wins a round iff no live neighbour is AHEAD of it in the order (priority desc, index asc); and whoever wins
// is ahead of all its live neighbours, so a vertex is excluded iff one of the vertices ahead of it won.  Both tests need one
// mask per vertex only - the neighbours ahead of it, a bit per vertex of the graph - built with ONE walk over the adjacency
// (lgs_mask_build); a round is two AND-OR sweeps over ceil(ng / 64) words against the graph's live / winners words and two
// barriers, whatever the degrees (lgs_mask_rounds).  Same synchronous rounds - same sets, same round counts - as lgs_rounds.
constexpr int kLgsMaskWords = 16;  // 1 024 vertices

// am[ng][W64] (zeroed here) <- the neighbours ahead of each vertex.  pr: priorities in LDS.  Barriers inside; the masks are
// complete when it returns.
template <int BLOCK>
__device__ __forceinline__ void lgs_mask_build(const int32_t* row_ptr, const int32_t* col_idx, int n0, int ng, const double* pr,
                                               unsigned long long* am, int W64) {
    for (int i = threadIdx.x; i < ng * W64; i += BLOCK) am[i] = 0ull;
    __syncthreads();
    // as many lanes per vertex as the workgroup affords (<= 8), sixteen neighbours in flight per lane (L2 round trips are
    // what this walk costs)
    int lsh = 0;
    while (lsh < 3 && (ng << (lsh + 1)) <= BLOCK) ++lsh;
    const int lpv = 1 << lsh;
    const int v = (int)threadIdx.x >> lsh, sub = (int)threadIdx.x & (lpv - 1);
    if (v < ng) {
        const double pv = pr[v];
        const int rs = row_ptr[n0 + v], re = row_ptr[n0 + v + 1];
        unsigned* row = reinterpret_cast<unsigned*>(am + (size_t)v * W64);
        constexpr int kFly = 16;
        for (int j = rs + sub; j < re; j += kFly * lpv) {
            int uu[kFly];
#pragma unroll
            for (int i = 0; i < kFly; ++i) uu[i] = (j + i * lpv < re) ? col_idx[j + i * lpv] - n0 : -1;
#pragma unroll
            for (int i = 0; i < kFly; ++i) {
                const int u = uu[i];
                if ((unsigned)u < (unsigned)ng) {  // (columns outside the graph are reported where the batch is validated)
                    const double pu = pr[u];
                    if ((pu > pv) || (pu == pv && u < v)) atomicOr(row + (u >> 5), 1u << (u & 31));
                }
            }
        }
    }
    __syncthreads();
}

// The rounds.  Thread tv stands for vertex tv; `my`: it takes part (tv < ng and not masked out); st[tv] is set to 1 / 2 when
// it joins / is excluded.  live / wonm: [kLgsMaskWords] words each (a word per wave).  Returns the number of rounds; every
// state byte is visible to the workgroup when it returns.
template <int BLOCK>
__device__ __forceinline__ int lgs_mask_rounds(int tv, bool my, const unsigned long long* am, int W64, unsigned long long* liveA,
                                               unsigned long long* liveB, unsigned long long* wonm, uint8_t* st, int max_rounds) {
    unsigned long long aw[kLgsMaskWords];
#pragma unroll
    for (int w = 0; w < kLgsMaskWords; ++w) aw[w] = (my && w < W64) ? am[(size_t)tv * W64 + w] : 0ull;
    const int wave = threadIdx.x >> 6;
    const bool lead = (threadIdx.x & 63) == 0 && wave < kLgsMaskWords;
    {
        const unsigned long long m0 = __ballot(my);
        if (lead) liveA[wave] = m0;
    }
    int rounds = 0;
    unsigned long long* lcur = liveA;
    unsigned long long* lnext = liveB;
    for (;;) {
        __syncthreads();  // this round's live words are written
        unsigned long long any = 0ull, t = 0ull;
#pragma unroll
        for (int w = 0; w < kLgsMaskWords; ++w) {
            if (w < W64) {
                const unsigned long long lw = lcur[w];
                any |= lw;
                t |= aw[w] & lw;
            }
        }
        if (any == 0ull || (max_rounds > 0 && rounds >= max_rounds)) break;
        ++rounds;
        const bool won = my && t == 0ull;
        {
            const unsigned long long wm = __ballot(won);
            if (lead) wonm[wave] = wm;
        }
        __syncthreads();
        unsigned long long k2 = 0ull;
#pragma unroll
        for (int w = 0; w < kLgsMaskWords; ++w)
            if (w < W64) k2 |= aw[w] & wonm[w];
        const bool killed = my && !won && k2 != 0ull;
        if (won) st[tv] = 1;
        else if (killed) st[tv] = 2;
        my = my && !won && !killed;
        {
            const unsigned long long m1 = __ballot(my);
            if (lead) lnext[wave] = m1;
        }
        unsigned long long* sw = lcur; lcur = lnext; lnext = sw;
    }
    __syncthreads();  // every state byte is written
    return rounds;
}

// The local greedy search's synchronous rounds (heuristics.py:77-116) on AHEAD lists: every undecided vertex's neighbours ahead of
// it in the order (priority desc, index asc) are compacted once to the front of its row of the 16-bit column lists in LDS; a round
// is then two walks over those - who has no undecided vertex ahead wins; who has a winner ahead (a winner beats all its undecided
// neighbours, so it is ahead of each of them) leaves - on state / flag BYTES only, half the entries, two barriers (lgs_rounds.h
// walks every neighbour with a float64 compare in the first phase, lets the winners push to all theirs in a second, updates in a
// third: three).  Same winners in the same rounds, hence the same states and the same round count.  A whole search only
// (max_rounds <= 0): a single round does not pay for the lists.  acnt: [ng] 16-bit counts; cl: WRITABLE column lists in LDS.
// No statistics (the _count / _stats / _overhead variants keep lgs_rounds).
// FLAG_OR: the votes through acc64[3] and acc64[2] in turn instead of __syncthreads_count (see lgs_rounds).
template <int BLOCK, bool FLAG_OR>
__device__ __forceinline__ int lgs_vote(bool mine, unsigned long long* acc64, int which) {
    if constexpr (FLAG_OR) {
        // (two words in turn: a thread may set this vote's word while a slower one still reads the last vote's)
        unsigned long long* word = acc64 + 2 + (which & 1);
        if (threadIdx.x == 0) acc64[2 + ((which + 1) & 1)] = 0;  // (the other word: everybody has read it - it decided to come here)
        if (mine) *word = 1;
        __syncthreads();
        return *word != 0 ? 1 : 0;
    } else {
        return __syncthreads_count(mine) ? 1 : 0;
    }
}
template <int BLOCK, bool FLAG_OR>
__device__ __forceinline__ int lgs_rounds_ahead(int ng, int e0, const double* pr, uint8_t* st, uint8_t* nw, uint16_t* cl, const int* rol,
                                                uint16_t* acnt, unsigned long long* acc64) {
    if (FLAG_OR) {
        if (threadIdx.x == 0) { acc64[2] = 0; acc64[3] = 0; }
        __syncthreads();
    }
    int votes = 0;
    int mine = 0;
    for (int v = threadIdx.x; v < ng; v += BLOCK) {
        int cnt = 0;
        if (st[v] == 0) {
            ++mine;
            const double pv = pr[v];
            const int rs = rol[v] - e0, re = rol[v + 1] - e0;
            for (int j = rs; j < re; ++j) {
                const int u = cl[j];
                if (u < ng && u != v && st[u] == 0) {
                    const double pu = pr[u];
                    if (pu > pv || (pu == pv && u < v)) cl[rs + cnt++] = (uint16_t)u;  // (behind the read position: same thread, in order)
                }
            }
        }
        acnt[v] = (uint16_t)cnt;
        nw[v] = 0;
    }
    int remaining = lgs_vote<BLOCK, FLAG_OR>(mine > 0, acc64, votes++);
    int rounds = 0;
    while (remaining) {
        for (int v = threadIdx.x; v < ng; v += BLOCK) {  // who wins this round: nobody undecided ahead
            if (st[v] != 0) continue;
            const int rs = rol[v] - e0, n = acnt[v];
            bool lost = false;
            int k = 0;
            for (; k + 3 < n; k += 4) {
                const int u0 = cl[rs + k], u1 = cl[rs + k + 1], u2 = cl[rs + k + 2], u3 = cl[rs + k + 3];
                lost |= (st[u0] == 0) | (st[u1] == 0) | (st[u2] == 0) | (st[u3] == 0);
            }
            for (; k < n; ++k) lost |= st[cl[rs + k]] == 0;
            nw[v] = lost ? 0 : 1;
        }
        __syncthreads();
        mine = 0;
        for (int v = threadIdx.x; v < ng; v += BLOCK) {  // winners join; who has a winner ahead leaves
            if (st[v] != 0) continue;
            if (nw[v]) { st[v] = 1; continue; }
            const int rs = rol[v] - e0, n = acnt[v];
            bool killed = false;
            int k = 0;
            for (; k + 3 < n; k += 4) {
                const int u0 = cl[rs + k], u1 = cl[rs + k + 1], u2 = cl[rs + k + 2], u3 = cl[rs + k + 3];
                killed |= (nw[u0] | nw[u1] | nw[u2] | nw[u3]) != 0;
            }
            for (; k < n; ++k) killed |= nw[cl[rs + k]] != 0;
            // (a flag left from an earlier round belongs to a member: whoever is adjacent to one has left in that round)
            if (killed) st[v] = 2; else ++mine;
        }
        remaining = lgs_vote<BLOCK, FLAG_OR>(mine > 0, acc64, votes++);
        ++rounds;
    }
    return rounds;
}


}  // namespace dgcn
