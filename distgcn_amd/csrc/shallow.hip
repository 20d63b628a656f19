// One-layer models (F -> 1: the paper's headline "shallow GCN", README.md:14; BASELINE configs C1, C2, C4 at l = 1)
// through dgcn_solve_batch: adjacency in, selected set out, ONE small workgroup per graph.
//
// k_fused serves these too, but it is built around a 32-wide LDS image (two 128-byte rows per vertex, entry records,
// row ranking for the wave-lockstep gather) that a width-1 layer never uses, and its greedy phase first turns the
// priorities into unique ranks - N^2 comparisons - which pays over 19 hidden layers and not here.  Measured on the
// C2 batch (500 ER N=100 graphs), hundred-cycle phase clocks of k_fused: row pointers 39, entries 40, row order 15,
// layer 46, priorities + ranks + rounds 175, tail 21 (profiles/r03_shallow_phase_clocks.txt) - the kernel is a chain of
// dependent global / LDS round trips, not a bandwidth problem.  This kernel is that chain and nothing else:
//
//   1. row_ptr[v], row_ptr[v + 1], weight[v] (one round trip; graph_ptr and the model's 2-4 numbers are scalar loads)
//   2. d^-1/2 table lookup by degree + the row's column ids, lpv lanes per row (second round trip)    -> LDS
//   3. entry values (float)(-(dinv[u] * dinv[v])) by all lanes, then ONE lane per row runs the contract's chain:
//      acc = fma((double)val_j, (double)z1[u_j], acc) over [diagonal, row entries in CSR order], score =
//      act((float)((double)z0 + acc [+ bias]))                         (gcn/layers.py:202-216; layer index 0: double chain)
//   4. priority (double)score * weight (mwis_dqn_call.py:232) into LDS, NaN check
//   5. local greedy rounds DIRECTLY on the float64 priorities (heuristics.py:77-116): a live vertex loses to a live
//      neighbour with (p_u > p_v) or (p_u == p_v and u < v); removed vertices hold NaN, which loses every comparison.
//      No ranking pass.  lpv lanes share a vertex's neighbour list.
//
// Same arithmetic contract, same outputs, same fault bits as k_fused (tests compare both with the twin bit for bit).
#include <atomic>

#include "common.h"

namespace dgcn {

struct ShallowArgs {
    const int32_t* graph_ptr;
    const int32_t* row_ptr;
    const int32_t* col_idx;
    const double* dinv_table;
    int32_t table_len;
    const float* X;  // [num_nodes][cin] or null
    float x_const;
    int32_t cin;
    const float* W;     // [cin][2]
    const float* bias;  // [1] or null
    int32_t act;
    const double* weights;
    int32_t predict_mwis;
    float* scores;
    uint8_t* state;
    int32_t* rounds;
    double* totals;
    int32_t* status;
    int32_t max_nodes;
    int32_t cap;  // entry slots per graph in LDS
    int32_t* done_flag;
    uint32_t* done_count;
    uint32_t done_target;
    unsigned long long* stamps;  // DGCN_DIAG builds only: [num_graphs][8] phase clocks of thread 0 (s_memtime)
};

#ifdef DGCN_DIAG
#define SH_STAMP(a, g, i, t0)                                                       \
    do {                                                                            \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();                 \
        if ((a).stamps && threadIdx.x == 0) (a).stamps[(size_t)(g) * 8 + (i)] = _t - (t0); \
        (t0) = _t;                                                                  \
    } while (0)
#else
#define SH_STAMP(a, g, i, t0) do { } while (0)
#endif

template <int BLOCK>
__device__ __forceinline__ bool sh_block_or(bool pred, unsigned* wflags) {
    if constexpr (BLOCK == 64) return __ballot(pred) != 0ull;
    const unsigned long long m = __ballot(pred);
    if ((threadIdx.x & 63) == 0) wflags[threadIdx.x >> 6] = m != 0ull;
    __syncthreads();
    unsigned any = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) any |= wflags[w];
    return any != 0;
}

template <int BLOCK>
__device__ __forceinline__ void sh_signal_done(const ShallowArgs& a) {
    if (!a.done_flag) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        const unsigned prev = atomicAdd(a.done_count, 1u);
        if (prev + 1u == a.done_target) __hip_atomic_store(a.done_flag, (int32_t)a.done_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// LONG: with the entry-parallel treatment of long rows compiled in (see `has_over` below): the host asks for it when
// the batch's largest graph has more than 128 vertices.  Smaller graphs cannot have rows much longer than what their
// lanes hold, and the extra uniform state costs the C2 launch 7 % (11.6 against 10.8 us, same box) even when unused.
// (1024-thread workgroups: two per CU = 8 waves per SIMD = at most 64 VGPRs and 96 SGPRs - left to itself the compiler
// takes 106 scalar registers and settles for 7)
template <int BLOCK, bool LONG>
__device__ __forceinline__ void shallow_body(const ShallowArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sh_raw[];
    const int g = blockIdx.x;
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1];
    const int ng = n1 - n0;
    // (slot max_nodes of dinv / pr / kr is a vertex nobody owns: d^-1/2 = 0, priority NaN - a lane's unused neighbour slots
    // point there, so the unrolled loops below need no bounds tests)
    const int mn1 = a.max_nodes + 1;
    double* dinv = reinterpret_cast<double*>(sh_raw);       // [max_nodes + 1]
    double* pr = dinv + mn1;                                 // [max_nodes + 1]
    double* red = pr + mn1;                                  // [16] block-sum partials
    float* z1 = reinterpret_cast<float*>(red + 16);          // [max_nodes + 1]
    float* vals = z1 + mn1;                                  // [cap]
    unsigned* wflags = reinterpret_cast<unsigned*>(vals + a.cap);              // [2][16] round votes, [32] NaN flag
    unsigned short* nbr = reinterpret_cast<unsigned short*>(wflags + 48);      // [cap] local neighbour ids
    unsigned short* kr = nbr + a.cap;                                          // [max_nodes + 1] round a vertex left in (0xFFFF: still in)
    if (ng <= 0) {
        if (threadIdx.x == 0) {
            if (a.rounds) a.rounds[g] = 0;
            if (a.totals) a.totals[g] = 0.0;
        }
        sh_signal_done<BLOCK>(a);
        return;
    }
    // lanes per vertex: as many as the block affords (<= 8)
    int lsh = 0;
    while (lsh < 3 && (ng << (lsh + 1)) <= BLOCK) ++lsh;
    const int lpv = 1 << lsh;
    const int vv = threadIdx.x >> lsh, sub = threadIdx.x & (lpv - 1);
    const bool mine = vv < ng;
    int fault = 0;
    unsigned long long tclk = 0;
#ifdef DGCN_DIAG
    tclk = __builtin_amdgcn_s_memtime();
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)g * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
    (void)tclk;
    // ---- round trip 1: row bounds, weight, the model's numbers, this vertex's features
    const int e0 = a.row_ptr[n0], e1 = a.row_ptr[n1];
    int rs = 0, re = 0;
    double w = 0.0;
    if (mine) {
        rs = a.row_ptr[n0 + vv];
        re = a.row_ptr[n0 + vv + 1];
        if (a.weights) w = a.weights[n0 + vv];
    }
    float z0 = 0.f, z1v = 0.f;
    for (int k = 0; k < a.cin; ++k) {  // layer 0's transform: float32 fmaf chain over the input features
        const float x = (a.X && mine) ? a.X[(size_t)(n0 + vv) * a.cin + k] : a.x_const;
        z0 = fmaf(x, a.W[k * 2 + 0], z0);
        z1v = fmaf(x, a.W[k * 2 + 1], z1v);
    }
    const float bias = a.bias ? a.bias[0] : 0.f;
    SH_STAMP(a, g, 0, tclk);
    // ---- round trip 2: degree table + the graph's columns
    const int deg = re - rs;
    double dv = 0.0;
    if (mine && sub == 0) {
        if (deg < a.table_len) dv = a.dinv_table[deg]; else fault |= DGCN_FAULT_DEGREE_RANGE;
    }
    // the graph's column ids, cooperatively and coalesced (thread t takes entries t, t + BLOCK, ..: eight loads in flight), as
    // 16-bit local ids; whose row an entry belongs to is not needed here (its position is its index)
    const bool graph_fits = (e1 - e0) <= a.cap;  // (the host sized `cap` for the largest graph: always true for a sane batch)
    const bool fits = mine && graph_fits;
    if (!graph_fits) fault |= DGCN_FAULT_BAD_COLUMN;
    if (graph_fits) {
        const int total = e1 - e0;
        for (int base = threadIdx.x; base < total; base += 8 * BLOCK) {
            int c[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) c[i] = (base + i * BLOCK < total) ? a.col_idx[e0 + base + i * BLOCK] : n0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (base + i * BLOCK < total) {
                    int u = c[i] - n0;
                    if (u < 0 || u >= ng) { fault |= DGCN_FAULT_BAD_COLUMN; u = 0xFFFF; }
                    nbr[base + i * BLOCK] = (unsigned short)u;
                }
            }
        }
    }
    SH_STAMP(a, g, 1, tclk);
    const int nobody = a.max_nodes;
    // Rows longer than the kNb * lpv entries their lanes keep in registers (BA hubs: up to ~150 entries on 2 lanes) would
    // walk the rest one entry per lane and round trip - 65 dependent trips per round for the worst graph of the C4 mix,
    // and the launch ends with its slowest graph.  A graph that has such rows (`has_over`, uniform in the workgroup) does
    // that work entry-parallel instead, all threads on all such entries: the entry values below, the comparisons of the
    // greedy rounds further down.  Graphs without (ER N = 100, p = 0.1: the C2 batch, whose longest rows are a few entries
    // over) run as before, barrier for barrier: "long" starts at kLongTrips lane trips past the registers.
    constexpr int kNb = 8;
    constexpr int kLongTrips = 16;  // (8: ER N = 200, p = 0.1 graphs - a row or two of 33+ entries on two lanes - lost 12 %: 26.1 against 23.3 us per launch)
    int* rowstart = reinterpret_cast<int*>(pr);  // [ng + 1] entry offsets of the rows (until the priorities are written)
    if (mine && sub == 0) {
        dinv[vv] = dv;
        z1[vv] = z1v;
        kr[vv] = 0xFFFFu;
        if constexpr (LONG) rowstart[vv] = rs - e0;
    }
    if constexpr (LONG) {  // "some row of this graph is long": one vote per wave (the round votes' first bank is free until round 2)
        const unsigned long long m = __ballot(mine && sub == 0 && deg > (kNb + kLongTrips) * lpv);
        if ((threadIdx.x & 63) == 0) wflags[threadIdx.x >> 6] = m != 0ull;
    }
    if (threadIdx.x == 0) {
        dinv[nobody] = 0.0;
        z1[nobody] = 0.f;
        pr[nobody] = __longlong_as_double(0x7ff8000000000000ll);
        kr[nobody] = 0;
        wflags[32] = 0u;
        if constexpr (LONG) { wflags[34] = 0u; rowstart[ng] = e1 - e0; }
    }
    __syncthreads();
    SH_STAMP(a, g, 2, tclk);
    // ---- entry values by all lanes of the row, then the chain by its first lane (LDS operations of one wave complete in
    // order and the lpv lanes of a row sit in one wave: no barrier between the writes and the reads).  A lane keeps the ids
    // of its first kNb entries in registers for the greedy rounds below.
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    const double dvv = mine ? dinv[vv] : 0.0;
    int nb[kNb];
#pragma unroll
    for (int i = 0; i < kNb; ++i) {
        const int j = rs + sub + i * lpv;
        int u = (fits && j < re) ? (int)nbr[j - e0] : nobody;
        if (u == 0xFFFF) u = nobody;
        nb[i] = u;
    }
    if (fits) {
        double dn[kNb];
#pragma unroll
        for (int i = 0; i < kNb; ++i) dn[i] = dinv[nb[i]];
#pragma unroll
        for (int i = 0; i < kNb; ++i) {
            const int j = rs + sub + i * lpv;
            if (nb[i] == vv) fault |= DGCN_FAULT_SELF_LOOP;
            if (j < re) vals[j - e0] = (float)(-(dn[i] * dvv));
        }
    }
    // (the votes are read here, behind the register path, so that their round trip is not on every graph's critical path)
    bool has_over = false;
    if constexpr (LONG) {
#pragma unroll
        for (int wv = 0; wv < BLOCK / 64; ++wv) has_over |= wflags[wv] != 0u;
        has_over = has_over && graph_fits;
    }
    if (has_over) {
        // the rest (and, harmlessly, the first kNb * lpv entries of every row once more) entry-parallel: a thread takes eight
        // consecutive entries, finds the row of the first by bisection over the row starts and walks on from there.  Same
        // expression, same bits as the row-parallel form.
        const int total = e1 - e0;
        for (int base = threadIdx.x * 8; base < total; base += BLOCK * 8) {
            int ids[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ids[i] = base + i < total ? (int)nbr[base + i] : 0xFFFF;
            int lo = 0, hi = ng;  // last row whose start is <= base (rows without entries are stepped over below)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (rowstart[mid] <= base) lo = mid; else hi = mid;
            }
            int v = lo, vend = rowstart[v + 1];
            double dvr = dinv[v];
#pragma unroll
            for (int h = 0; h < 8; h += 4) {  // (four at a time: the 1024-thread variant has 64 registers)
                double du[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) du[i] = dinv[ids[h + i] == 0xFFFF ? nobody : ids[h + i]];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = base + h + i;
                    if (j < total) {
                        while (j >= vend) { ++v; vend = rowstart[v + 1]; dvr = dinv[v]; }
                        if (ids[h + i] == v) fault |= DGCN_FAULT_SELF_LOOP;
                        vals[j] = ids[h + i] == 0xFFFF ? 0.f : (float)(-(du[i] * dvr));
                    }
                }
            }
        }
        __syncthreads();
    } else if (fits) {
#pragma unroll 4
        for (int j = rs + sub + kNb * lpv; j < re; j += lpv) {  // (at most kLongTrips trips)
            const int u = nbr[j - e0];
            if (u == vv) fault |= DGCN_FAULT_SELF_LOOP;
            vals[j - e0] = u == 0xFFFF ? 0.f : (float)(-(dinv[u] * dvv));
        }
    }
    float score = 0.f;
    double p = 0.0;
    if (mine && sub == 0) {
        double acc = fma(1.0, (double)z1v, 0.0);  // the diagonal entry of L comes first
        if (fits) {
            int j = rs;
            if (a.X) {
#pragma unroll 4
                for (; j < re; ++j) {
                    const int u = nbr[j - e0];
                    acc = fma((double)vals[j - e0], (double)z1[u == 0xFFFF ? nobody : u], acc);
                }
            } else {  // constant features: every z1[u] is this vertex's own z1
                const double zd = (double)z1v;
                for (; j + 8 <= re; j += 8) {  // eight independent reads in flight, chain order unchanged
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = vals[j + i - e0];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc = fma((double)v[i], zd, acc);
                }
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = j + i < re ? vals[j + i - e0] : 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) if (j + i < re) acc = fma((double)v[i], zd, acc);
            }
        }
        acc = (double)z0 + acc;
        if (a.bias) acc += (double)bias;
        score = apply_act((float)acc, a.act);
        if (a.scores) a.scores[n0 + vv] = score;
        p = (double)score;
        if (a.predict_mwis && a.weights) p *= w;
        if (p != p) wflags[32] = 1u;
        pr[vv] = p;
    }
    __syncthreads();  // priorities (and the NaN flag) are in LDS
    if (wflags[32]) {
        if (threadIdx.x == 0) {
            atomicOr(a.status, fault | DGCN_FAULT_NAN_PRIORITY);
            if (a.rounds) a.rounds[g] = -1;
            if (a.totals) a.totals[g] = 0.0;
        }
        if (mine && sub == 0) a.state[n0 + vv] = 0;
        sh_signal_done<BLOCK>(a);
        return;
    }
    SH_STAMP(a, g, 3, tclk);
    // ---- local greedy rounds on the priorities themselves.  kr[u] = the round u left the graph in (winner or removed
    // neighbour), 0xFFFF while it is in: in round r a vertex counts as present iff kr >= r, so what a fast wave removes in
    // round r does not change what a slow wave still reads in round r - ONE barrier per round (which also carries the
    // "anybody still in?" votes) instead of two.  Priorities never change.
    // (Tried: sixteen register neighbours per lane in the long-row kernels - ids packed two to a register, compared four at a
    // time to stay inside 64 registers -, which cuts the listed entries of the densest graphs to a quarter: 43.4 - 43.6 against
    // 41.2 - 41.4 us for the C4 share, the serialised compares cost more than the shorter sweeps save.  Weight and priority
    // fetched again for the totals instead of carried through the rounds (4 registers): inside the noise on C2, + 1 % on C4.)
    // (Tried: priority and stamp of a vertex side by side in one 16-byte word, ONE ds_read_b128 per neighbour instead of a
    // ds_read_b64 and a ds_read_u16: 10.8 - 11.8 against 10.7 - 11.3 us for the C2 launch on one box - the round is a chain of
    // round trips, not a count of LDS instructions.)
    // (Tried: ONE WAVE per graph with ceil(N / 64) vertices per lane and no barrier at all.  27.9 us against 12.6 us for
    // the C2 launch, 144 against 53 us on the BA mix: the phases are chains of dependent LDS round trips, and one wave has
    // nothing to issue while it waits - several waves per graph hide each other's latency.)
    // Graphs with long rows: the entries past a row's first kNb * lpv as one flat list of (row << 16 | neighbour) words
    // (in the space of the entry values, which nobody reads any more), compared entry-parallel in every round; a row that
    // loses through one of them is stamped in `lostr` (in the space of z1), read back after one more barrier; a winner is
    // stamped in `wonr`, and a third sweep-and-barrier takes its listed neighbours out.
    unsigned* ov = reinterpret_cast<unsigned*>(vals);
    unsigned short* lostr = reinterpret_cast<unsigned short*>(z1);  // [max_nodes + 1] round in which a row last lost through the list
    unsigned short* wonr = lostr + mn1;                             // [max_nodes + 1] round in which a row won
    int ovn = 0;
    if (has_over) {
        const int over = fits ? max(0, deg - kNb * lpv) : 0;
        int pos = 0;
        if (sub == 0 && over > 0) pos = (int)atomicAdd(&wflags[34], (unsigned)over);
        pos = __shfl(pos, (int)(threadIdx.x & 63u) & ~(lpv - 1));
        if (mine && sub == 0) { lostr[vv] = 0; wonr[vv] = 0; }
        for (int j = rs + sub + kNb * lpv; j < re; j += lpv) {
            const int u0 = nbr[j - e0];
            ov[pos + (j - rs - kNb * lpv)] = ((unsigned)vv << 16) | (unsigned)(u0 == 0xFFFF ? nobody : u0);
        }
        __syncthreads();
        ovn = (int)wflags[34];
    }
    int rounds = 0;
    bool member = false;
    for (unsigned r = 1;; ++r) {
        const double pv = mine ? pr[vv] : qnan;
        const bool live = mine && (unsigned)kr[vv] >= r;
        bool lost = false;
        // which register neighbours were in at the start of this round, for the winners' stamps: the stamps themselves, or
        // (LONG: the 1024-thread variant has 64 registers) one bit each.  (The bits cost the C2 launch 0.9 us of 10.6.)
        unsigned pres = 0u;
        unsigned kuk[kNb];
        (void)pres; (void)kuk;
        if (live && fits) {
            double pu[kNb];
            unsigned ku[kNb];
#pragma unroll
            for (int i = 0; i < kNb; ++i) { pu[i] = pr[nb[i]]; ku[i] = kr[nb[i]]; }
#pragma unroll
            for (int i = 0; i < kNb; ++i) {
                if constexpr (LONG) pres |= (unsigned)(ku[i] >= r) << i;
                else kuk[i] = ku[i];
                lost |= (ku[i] >= r) & ((pu[i] > pv) | ((pu[i] == pv) & (nb[i] < vv)));
            }
            if (!has_over) {
#pragma unroll 4
                for (int j = rs + sub + kNb * lpv; j < re; j += lpv) {  // (at most kLongTrips trips)
                    const int u0 = nbr[j - e0];
                    const int u = u0 == 0xFFFF ? nobody : u0;
                    const double q = pr[u];
                    lost |= ((unsigned)kr[u] >= r) & ((q > pv) | ((q == pv) & (u < vv)));
                }
            }
        }
        if (has_over) {
#pragma unroll 4
            for (int k = threadIdx.x; k < ovn; k += BLOCK) {
                const unsigned e = ov[k];
                const int v = (int)(e >> 16), u = (int)(e & 0xffffu);
                if ((unsigned)kr[v] >= r && (unsigned)kr[u] >= r) {
                    const double q = pr[u], pq = pr[v];
                    if ((q > pq) | ((q == pq) & (u < v))) lostr[v] = (unsigned short)r;
                }
            }
            __syncthreads();
            lost |= mine && (unsigned)lostr[vv] == r;
        }
        for (int off = 1; off < lpv; off <<= 1) lost |= (bool)__shfl_xor((int)lost, off);
        const bool won = live && !lost;
        {
            const unsigned long long m = __ballot(live);
            if ((threadIdx.x & 63) == 0) wflags[(r & 1) * 16 + (threadIdx.x >> 6)] = m != 0ull;
        }
        if (won) {  // (only vertices still in are stamped: a stale neighbour must not look present again to this round's readers)
            if (fits) {
#pragma unroll
                for (int i = 0; i < kNb; ++i)
                    if ((LONG ? ((pres >> i) & 1u) != 0u : kuk[i] >= r) && nb[i] != nobody) kr[nb[i]] = (unsigned short)r;
                if (!has_over)
                    for (int j = rs + sub + kNb * lpv; j < re; j += lpv) {  // (at most kLongTrips trips)
                        const int u = nbr[j - e0];
                        if (u != 0xFFFF && (unsigned)kr[u] >= r) kr[u] = (unsigned short)r;
                    }
            }
            if (has_over && sub == 0) wonr[vv] = (unsigned short)r;  // its listed neighbours leave below
            if (sub == 0) { kr[vv] = (unsigned short)r; member = true; }
        }
        if (has_over) {  // the winners' listed neighbours, entry-parallel (same rule: only vertices still in are stamped)
            __syncthreads();
#pragma unroll 4
            for (int k = threadIdx.x; k < ovn; k += BLOCK) {
                const unsigned e = ov[k];
                const int v = (int)(e >> 16), u = (int)(e & 0xffffu);
                if ((unsigned)wonr[v] == r && u != nobody && (unsigned)kr[u] >= r) kr[u] = (unsigned short)r;
            }
        }
        __syncthreads();
        unsigned any = 0;
#pragma unroll
        for (int wv = 0; wv < BLOCK / 64; ++wv) any |= wflags[(r & 1) * 16 + wv];
        if (!any) break;  // nobody was in at the start of round r: it did not happen
        ++rounds;
    }
    SH_STAMP(a, g, 4, tclk);
    if (mine && sub == 0) {
        a.state[n0 + vv] = member ? 1 : 2;  // (the loop ends when nobody is in: a vertex either joined or was removed)
    }
    if (threadIdx.x == 0 && a.rounds) a.rounds[g] = rounds;
    if (a.totals) {
        // block-wide sum in a fixed order: shuffle tree inside a wave, then the wave partials in wave order
        double part = (mine && sub == 0 && member) ? (a.weights ? w : p) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        if constexpr (BLOCK > 64) {
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
            __syncthreads();
            part = red[0];
#pragma unroll
            for (int wv = 1; wv < BLOCK / 64; ++wv) part += red[wv];
        }
        if (threadIdx.x == 0) a.totals[g] = part;
    }
    if (fault) atomicOr(a.status, fault);
    sh_signal_done<BLOCK>(a);
    SH_STAMP(a, g, 5, tclk);
#ifdef DGCN_DIAG
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)g * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
}

// The kernels proper.  1024-thread workgroups: two per CU = 8 waves per SIMD = at most 64 VGPRs and 96 SGPRs - left to
// itself the compiler takes 106 scalar registers and settles for 7; the smaller shapes are better off without the limit
// (C2: 11.5 against 12.1 us with it).
template <int BLOCK, bool LONG>
__global__ __launch_bounds__(BLOCK) void k_shallow(ShallowArgs a) { shallow_body<BLOCK, LONG>(a); }
template <bool LONG>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_shallow_1024(ShallowArgs a) { shallow_body<1024, LONG>(a); }

static size_t shallow_lds(int max_nodes, int cap) {
    return (size_t)(max_nodes + 1) * (8 + 8 + 4 + 2) + 16 * 8 + 48 * 4 + (size_t)cap * 6 + 64;
}

// does this (batch, model) go through k_shallow?  One layer F -> 1 with two supports, graphs of <= 512 vertices whose
// neighbour lists fit the LDS.  Option "shallow" = 0 sends everything to k_fused (tests compare the two).
bool shallow_takes(const DgcnBatch* b, const DgcnModel* m) {
    if (opt(OPT_SHALLOW) == 0) return false;
    if (!m->layers_host || m->num_layers != 1 || m->num_supports != 2) return false;
    const DgcnLayer& L = m->layers_host[0];
    if (L.out_dim != 1 || L.in_dim < 1 || L.in_dim > 64) return false;
    if (b->max_nodes > 512) return false;
    const int cap = (b->max_graph_edges + 7) & ~7;
    return shallow_lds(max(b->max_nodes, 64), cap) <= 96 * 1024;
}

template <int BLOCK, bool LONG>
static int shallow_launch_bl(ShallowArgs& a, int B, size_t lds, hipStream_t s) {
    if (lds > 64 * 1024) {
        static std::atomic<int> raised[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!raised[dev & 63].load(std::memory_order_relaxed)) {
            const void* kern;
            if constexpr (BLOCK == 1024) kern = reinterpret_cast<const void*>(&k_shallow_1024<LONG>);
            else kern = reinterpret_cast<const void*>(&k_shallow<BLOCK, LONG>);
            if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
                return fail(DGCN_ERR_LAUNCH, "k_shallow: cannot reserve %zu bytes of LDS", lds);
            raised[dev & 63].store(1, std::memory_order_relaxed);
        }
    }
    TimedLaunch t("fused_solve", s);  // (the same timing family as k_fused: it is the same entry point's launch)
    if constexpr (BLOCK == 1024) DGCN_LAUNCH(t, (k_shallow_1024<LONG>), dim3(B), dim3(BLOCK), lds, s, a);
    else DGCN_LAUNCH(t, (k_shallow<BLOCK, LONG>), dim3(B), dim3(BLOCK), lds, s, a);
    return check_launch("k_shallow");
}

// The long-row treatment is compiled into its own kernels, for batches that can have long rows at all: largest graph above
// 128 vertices AND at least 24 entries per vertex in the densest one (the BA mix: 37; ER N = 200, p = 0.1: 22).  Carried
// along unused it costs 6 - 7 % (ER N = 200: 24.5 against 23.0 us; C2: 11.6 against 10.7) - scalar registers spilled to lanes.
// Either kernel is correct for every batch: the choice is about time only.
template <int BLOCK>
static int shallow_launch_b(ShallowArgs& a, int B, size_t lds, hipStream_t s) {
    if constexpr (BLOCK >= 512) {
        bool lng = (long)a.cap >= 24L * a.max_nodes;
        if (const int want = opt(OPT_SHALLOW_LONG); want >= 0) lng = want != 0;  // tuning / tests
        if (lng) return shallow_launch_bl<BLOCK, true>(a, B, lds, s);
    }
    return shallow_launch_bl<BLOCK, false>(a, B, lds, s);
}

int shallow_solve(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, const float* X,
                  float x_const, const double* weights, int32_t predict_mwis, float* scores, uint8_t* state, int32_t* rounds,
                  double* totals, int32_t* status, const DoneHook& hook, hipStream_t s) {
    const DgcnLayer& L = m->layers_host[0];
    ShallowArgs a = {};
    a.graph_ptr = b->graph_ptr;
    a.row_ptr = b->row_ptr;
    a.col_idx = b->col_idx;
    a.dinv_table = dinv_table;
    a.table_len = table_len;
    a.X = X;
    a.x_const = x_const;
    a.cin = L.in_dim;
    a.W = L.weights;
    a.bias = L.bias;
    a.act = L.act;
    a.weights = weights;
    a.predict_mwis = predict_mwis;
    a.scores = scores;
    a.state = state;
    a.rounds = rounds;
    a.totals = totals;
    a.status = status;
    a.max_nodes = max(b->max_nodes, 64);
    a.cap = (b->max_graph_edges + 7) & ~7;
    a.done_flag = hook.flag;
    a.done_count = hook.count;
    a.done_target = hook.target;
#ifdef DGCN_DIAG
    a.stamps = reinterpret_cast<unsigned long long*>(static_cast<uintptr_t>(opt64(OPT_DIAG_STAMPS)));
#endif
    const size_t lds = shallow_lds(a.max_nodes, a.cap);
    // threads: a vertex per thread at least; small graphs get several lanes per vertex (<= 8) out of a 64..256-thread block
    const int mn = b->max_nodes;
    // measured (tools/runs/r03_gpu9.sh, 500 graphs per launch): ER N=100: 17.9 / 12.3 / 13.3 / 24.4 us with 128 / 256 / 512 / 1024
    // threads; ER N=200: 22.8 (512) / 27.8 (1024); BA mix up to N=300 (hub rows): 53.0 (512) / 44.8 (1024)
    int block = mn <= 16 ? 64 : mn <= 64 ? 128 : mn <= 128 ? 256 : mn <= 256 ? 512 : 1024;
    if (const int want = opt(OPT_SHALLOW_BLOCK); want > 0) {  // tuning: any of 64 / 128 / 256 / 512 / 1024 that holds a vertex per thread
        if ((want == 64 || want == 128 || want == 256 || want == 512 || want == 1024) && want >= mn) block = want;
    }
    if (block == 64) return shallow_launch_b<64>(a, b->num_graphs, lds, s);
    if (block == 128) return shallow_launch_b<128>(a, b->num_graphs, lds, s);
    if (block == 256) return shallow_launch_b<256>(a, b->num_graphs, lds, s);
    if (block == 512) return shallow_launch_b<512>(a, b->num_graphs, lds, s);
    return shallow_launch_b<1024>(a, b->num_graphs, lds, s);
}

}  // namespace dgcn
