// Fused per-graph path: one workgroup owns one graph from the first layer to the selected set.
//
//   dgcn_gcn_forward_batch(mode = 1)  : all GraphConvolution layers (gcn/models.py:536-573), scores out
//   dgcn_solve_batch                  : support construction (gcn/utils.py:120-127, 258-274) + all layers +
//                                       priority product (mwis_dqn_call.py:232) + local greedy search
//                                       (heuristics.py:77-116) in ONE launch: adjacency in, set out
//
// Why: at N ~ 200 one graph's working set fits the 160 KB LDS of a CU.  The layer-by-layer path moves
// ~1.66 MB per graph through HBM/L2 over 20 layers and pays ~40 kernel boundaries; here HBM sees the
// adjacency once (nnz*4 + N*12 bytes in) and the membership once (N bytes out).
//
// LDS image of a graph (N <= 512 vertices, hidden width 32):
//   bufA[N][32] f32   H, then Z0 = H.W0 in place, then H' in place        (16-byte chunks XOR-swizzled by row)
//   bufB[N][32] f32   Z1 = H.W1, the operand every neighbour gather reads  (chunks swizzled inside each
//                     64-byte half by row & 3: keeps the half-row bank windows the random gather
//                     conflicts least on, and lets the MFMA epilogue store 16-byte chunks 2-way instead of 8-way)
//   vals[] f32, words[] u16, rinfo[N] u32, perm[N] u16   the support L = I - D^-1/2 A D^-1/2, diagonal
//                     first; word = (u << 7) | ((u & 3) << 4), a lane's gather address is word ^ (chunk << 4);
//                     every row's entry list starts
//                     at an even slot so two words / two values come with one 4- / 8-byte LDS read;
//                     rinfo = start | count << 16; perm = rows by descending entry count, dealt to the
//                     waves in snake order so the 8 rows a wave walks in lockstep have equal length
//   (global scratch) the same entries once more as 8-byte records {value, gather word}, block-major and padded to the trip
//                     count of each 16-row block (row_blocks_init): what the 32-wide aggregations walk
// N = 200, nnzL ~ 4.2k -> ~79 KB: two workgroups per CU.  (Measured in round 3, tools/ablate_fused.py, one against two per
// CU: the two hide each other's image build and greedy rounds almost completely, each other's layer phases only by a fifth -
// fp32 MFMA keeps the other waves' vector instructions out of the SIMD.)
// The residual-graph variant (MASKED) builds this image on the vertices that are still undecided, renumbered 0 .. na - 1.
//
// Arithmetic is the library-wide contract (include/dgcn.h): transform = k-ordered fmaf chain (fp32 MFMA
// 16x16x4 is exactly that), aggregate = fmaf chain in entry order from 0, then Z0 + sum, + bias,
// activation; the aggregation of layer index 0 and the transform of layer index 1 carry their chains in
// double and round once (v_mfma_f64_16x16x4_f64 is exactly the ascending fma chain: tools/micro/mfma_f64.hip).
// Scores are therefore bit-identical to mode 0 and to oracle/dgcn_oracle.c.
//
// Shapes handled here: first layer any width -> 32 (VALU), hidden layers 32 -> 32 (MFMA), last layer
// -> 1; or a single layer F -> 1.  Anything else returns DGCN_ERR_UNSUPPORTED and the caller uses mode 0.
#include <atomic>
#include <random>

#include "common.h"
#include "tile_ops.h"
#include "wave_reduce.h"
#include <type_traits>

namespace dgcn {

constexpr int kMaxFusedLayers = 64;
constexpr int kFusedBlock = 512;      // threads per workgroup when two or more graphs share a CU
constexpr int kFusedBlockBig = 1024;  // ... when one graph's image takes more than half the LDS (it has the CU to itself)
constexpr int kFusedMaxNodes = 512;
// k_fused<.., GW>: where the gather words sit in LDS before the layers (inside bufA, behind P0's row starts and the compact
// form's wave sums) and behind them (inside bufB, behind the z1 scalars / priorities / 512 reduction slots)
constexpr int kGwWordsP0 = 2304;
constexpr int kGwWordsTail = 12288;
constexpr int kGwMaxNodes = 384;  // three row blocks of sixteen rows on each of eight waves

struct FusedLayer {
    const float* W;     // [cin][2*cout]
    const float* bias;  // [cout] or null
    int32_t cin, cout, act, pad;
};

struct FusedArgs {
    const int32_t* graph_ptr;
    const int32_t* row_ptr;
    const int32_t* col_idx;
    const float* vals;         // values of the given support CSR; unused when from_adj
    uint2* grec;               // [num_graphs][rec_cap] entry records {value bits, gather word} for the hidden aggregation
    float* gvals;              // k_fused<*, true>: [num_graphs][meta_cap] entry values kept in global memory
    unsigned short* gwords;    // k_fused<.., GW>: [num_graphs][meta_cap] gather words while the layers run
    const double* dinv_table;  // from_adj: float64 d^-1/2 table
    int32_t table_len;
    int32_t from_adj;          // CSR is the adjacency: build L (diagonal first) on the fly
    const float* X;            // [num_nodes][cin0] or null
    float x_const;
    float* scores;             // [num_nodes] or null
    const double* weights;     // vertex weights (priority product and totals) or null
    int32_t predict_mwis;
    int32_t do_lgs;
    uint8_t* state;
    int32_t* rounds;
    double* totals;
    int32_t* status;
    int32_t num_layers;
    int32_t max_nodes;
    int32_t meta_cap;
    int32_t rec_cap;           // records per graph in grec: block-major and padded (row_blocks_init), or row-major (cluster variant)
    int32_t prio_second;
    int32_t gw;                // 1: k_fused<.., GW> runs this launch (values and gather words in global scratch, two workgroups per CU)
    int32_t prio_gather;  // issue priority added during the aggregation phase (0..2)
    int32_t wide_passes;  // > 1: a two-layer stack F -> c -> 1 with 32 < c <= 32 * wide_passes: layers[0..P-1] are the
                          // first layer cut into 32-column blocks, layers[P] is the last layer (see fused_prepare)
    int32_t flags_off;  // byte offset of the block-OR scratch words inside the dynamic LDS; a 128-byte zero row follows
    // residual-graph variant (k_fused<true>): `state` is in/out, vertices with state != 0 are not part of the graph
    int32_t feature_mode;  // 1: X[v][*] = (float)(w[v] / (max residual w + 1e-9)), computed here
    int32_t greedy_mode;   // 0 local greedy rounds, 1 one centralised step (global best joins), 2 one rollout step
    int32_t max_rounds;    // greedy_mode 0: stop after this many rounds (0 = until every vertex is decided)
    int32_t beam;          // greedy_mode 2: number of candidates
    int32_t options;       // DGCN_RESIDUAL_* bits
    int32_t* progress;     // += 1 per graph that decided at least one vertex in this launch
    unsigned long long* tail_word;  // residual-graph variant, or null: atomicMax(tail_tag | undecided vertices of an active graph) -
    unsigned long long tail_tag;    // what tail.hip reads to see whether EVERY graph of the batch is small enough for it
    const int32_t* order;  // null, or the graph of workgroup i (largest graphs first: k_graph_rank)
    const int32_t* cedge;          // compact batch (common.h CompactHook), or null: first entry of every graph,
    const unsigned short* cdeg;    // entries per row,
    const unsigned short* ccol;    // column ids local to their graph (row_ptr / col_idx are null then)
    int32_t* done_flag;    // see DoneHook (common.h); null = no completion word
    uint32_t* done_count;
    uint32_t done_target;
    // cluster variant (k_fused<false, *, 512, true>): `cluster` workgroups per graph, see cluster_pull_rows()
    int32_t cluster;
    int32_t cluster_inject;  // test hook (option "test_cluster_fault"): report a placement fault although there is none
    unsigned long long nonce;  // this launch's value of the progress words ("my exchange rows are marked unwritten")
    int32_t num_graphs;
    float* xz;          // [num_graphs][3][max_nodes][32] Z1 rows on their way between the workgroups of a graph (slice l % 3)
    float* xs;          // [num_graphs][2][max_nodes] last layer: z1 scalars, then scores
    unsigned long long* xflag;  // G = num_graphs rounded up to 8: [G][8] 64-bit workgroup progress words, then [G][8] int32 XCC ids
    int32_t diag;  // DGCN_DIAG builds only: bit0 skip gathers, bit1 skip transforms, bit2 skip greedy rounds
    unsigned long long* stamps;  // DGCN_DIAG builds only: [num_graphs][16] wave-0 phase clocks (s_memtime)
    FusedLayer layers[kMaxFusedLayers];
};

#ifdef DGCN_DIAG
#define DIAG_ON(a, bit) (((a).diag >> (bit)) & 1)
// phase clock of workgroup g, slot i: accumulated (not overwritten) so per-layer phases sum up
// (kept in registers - the clock is a scalar - and written out once by STAMP_FLUSH: a global read-modify-write per phase,
// as in round 3, put a global round trip into every phase it measured)
#define STAMP(a, g, i, t0)                                                                    \
    do {                                                                                      \
        const unsigned long long _t = __builtin_amdgcn_s_memtime();                           \
        stamp_acc[i] += _t - (t0);                                                            \
        (t0) = _t;                                                                            \
    } while (0)
#define STAMP_FLUSH(a, g)                                                                     \
    do {                                                                                      \
        if ((a).stamps && stamp_wg && threadIdx.x == 0) {                                     \
            _Pragma("unroll") for (int _i = 0; _i < 12; ++_i) (a).stamps[(size_t)(g) * 64 + _i] += stamp_acc[_i]; \
            (a).stamps[(size_t)(g) * 64 + 15] = __builtin_amdgcn_s_memrealtime(); /* (slot 14: the same clock at the start) */ \
        }                                                                                     \
    } while (0)
#else
#define DIAG_ON(a, bit) 0
#define STAMP(a, g, i, t0) do { } while (0)
#define STAMP_FLUSH(a, g) do { } while (0)
#endif

// ---- first layer (input from global X or a constant): one thread per vertex, outputs in chunks of 16
__device__ __forceinline__ void set_prio(int p) {  // s_setprio takes an immediate
    if (p <= 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}

template <int BLOCK>
__device__ __forceinline__ void first_layer_transform(const FusedArgs& a, const FusedLayer& L, int n0, int ng,
                                                      float* bufA, float* bufB, float xfill) {
    const int cin = L.cin, ctot = 2 * L.cout;  // cout == kHid here
    for (int v = threadIdx.x; v < ng; v += BLOCK) {
        for (int c0 = 0; c0 < ctot; c0 += 16) {
            float acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            for (int k = 0; k < cin; ++k) {
                const float x = a.X ? a.X[(size_t)(n0 + v) * cin + k] : xfill;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = fmaf(x, L.W[k * ctot + c0 + i], acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
                float* dst = c0 < kHid ? bufA + swz(v, c0 + i) : bufB + swzB(v, c0 - kHid + i);
                *reinterpret_cast<float4*>(dst) = make_float4(acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
            }
        }
    }
}

// Z1 row chunk at ABSOLUTE LDS byte address `addr` (bufB starts at LDS offset 0).
typedef __attribute__((address_space(3))) const f32x4 lds_cf4;
__device__ __forceinline__ float4 lds_chunk(unsigned addr) {
    const f32x4 z = *reinterpret_cast<lds_cf4*>(addr);
    return make_float4(z[0], z[1], z[2], z[3]);
}

// The hidden aggregation's entry records, block-major (round 3).  The rows of a graph go through the aggregation in
// blocks of 16 (`perm` order: descending entry count), one block per wave pass, four entries of every row per trip.  A
// block of T trips owns 64 T records: record 64 t + lane is entry 4 t + (lane & 3) of the row in slot lane >> 2, or the
// neutral record {-0.0f, zero row} past the row's end - fmaf(-0.0f, +0.0f, acc) == acc for every acc, so a short row
// rides along to the block's trip count (that of its first row) without a compare, and the last 1..3 entries of a row
// need no code of their own.  What this buys: a trip's record load is 512 consecutive bytes behind a wave-uniform
// base, the trip count sits in a scalar register, and the per-lane bookkeeping of the row-major records (entry index,
// row end, compare, exec mask: 6 of 42 vector instructions per trip, on a phase that is paced by vector issue) is gone.
// The wave that gathers a block writes its records itself (same lane, same address: no barrier between).
constexpr int kMaxRowBlocks = 2;  // per wave: 256 vertices on 8 waves, 512 on 16 (larger graphs never get 512-thread workgroups)
struct RowBlock {
    int v;            // row of this lane's slot in the block, -1 = none
    int trips;        // trips of the block (wave-uniform), 0 = no such block
    unsigned base;    // first record of the block (wave-uniform)
    unsigned fx, fy;  // this lane's record of trip 0 {value bits, gather word}
};
struct RowBlocks {  // (named members, not an array: the kernel has no scratch memory, and an array that is not split
    RowBlock b0, b1;  // into registers early enough would sit there)
    RowBlock b2;      // (NB = 3 only - k_fused<.., GW>: 512-thread workgroups on graphs of up to 384 vertices; never touched otherwise)
    template <int K> __device__ __forceinline__ RowBlock& at() { if constexpr (K == 0) return b0; else if constexpr (K == 1) return b1; else return b2; }
    template <int K> __device__ __forceinline__ const RowBlock& at() const { if constexpr (K == 0) return b0; else if constexpr (K == 1) return b1; else return b2; }
};

template <int BLOCK, int NB = 2>
__device__ __forceinline__ bool row_blocks_init(RowBlocks& rb, int ng, const unsigned* rinfo, const unsigned short* perm,
                                                const float* vals, const unsigned short* words, uint2* rec, unsigned zrow,
                                                int rec_cap, bool diag_parity = false) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane >> 2, kq = lane & 3;
    // (DGCN_DIAG builds, DGCN_FUSED_DIAG bit 4: every neighbour's parity forced to what makes its 16-lane bank group
    // conflict-free - wrong results, a timing experiment: what would the aggregation cost without LDS bank conflicts?)
    auto fix_word = [&](unsigned w) -> unsigned {
        if (!diag_parity) return w;
        const int u = (int)(w >> 7);
        int u2 = (u & ~1) | ((s >> 2) & 1);
        if (u2 >= ng) u2 = u;
        return (unsigned)enc_word(u2);
    };
    constexpr int kWaves = BLOCK / 64;
    const int blocks = (ng + 15) >> 4;  // <= 32 (NB * kWaves)
    // trips of block `lane` and the records in front of it (every wave computes the same table: one scan, once per graph)
    // (at least one trip for a block that exists: rows without a single entry - removed vertices of a residual graph, empty
    // rows of a caller's support matrix - still get their Z0 (+ bias) through the activation)
    const int tl = lane < blocks ? max(1, (int)(((rinfo[perm[lane * 16]] >> 16) + 3) >> 2)) : 0;
    int incl = tl;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    // (Tried in round 3 and dropped: dealing the rows so that every wave gets a full first pass from the head of the
    // descending order plus an equal share of the short rows - an N = 200 graph's 13 blocks otherwise leave three waves
    // with one pass and five with two.  208 us against 200 us per C3 launch: the waves that finish early are what lets the
    // co-resident workgroup's transform through.  Also: the second workgroup of a CU handing its left-over tiles / row blocks
    // - the 9th .. 13th of an N = 200 graph - to waves one further on, so that the SIMD with four tiles instead of three is
    // not the same one for both: 200.4 - 201.0 against 201.0 - 202.0 us in the same build, inside the noise.)
    // The slice holds what fused_rec_cap() proves for a graph whose longest row has at most N entries.  A row longer than
    // that (repeated columns in the caller's matrix) could outgrow it: such a graph gets no records - and a fault bit from
    // the caller of this function - instead of writing past its slice.
    const int total = blocks > 0 ? __shfl(incl, blocks - 1) : 0;
    const bool fits_slice = total * 64 + 192 <= rec_cap;
    const uint2 nothing = make_uint2(0x80000000u, zrow);
    // (one call per block instead of an unrolled loop: with the record loop inside it the loop is unrolled only after the
    // pass that turns `rb` into registers has run, and the kernel would keep it in scratch memory)
    auto one_block = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const int blk = k * kWaves + ((k & 1) ? (kWaves - 1 - wave) : wave);  // (snake: 0 .. W-1, 2W-1 .. W, 2W .. 3W-1)
        const int src = blk < blocks ? blk : 0;
        const int trips = (blk < blocks && fits_slice) ? __shfl(tl, src) : 0;
        const unsigned base = (unsigned)(__shfl(incl, src) - __shfl(tl, src)) * 64u;
        RowBlock& B = rb.template at<k>();
        B.trips = __builtin_amdgcn_readfirstlane(trips);
        B.base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
        const int slot = blk * 16 + s;
        const bool has = blk < blocks && slot < ng;
        B.v = has ? (int)perm[slot] : -1;
        const unsigned ri = has ? rinfo[B.v] : 0u;
        const int start = ri & 0xffff, cnt = ri >> 16;
        uint2 r0 = nothing;
        if (kq < cnt) r0 = make_uint2(__float_as_uint(vals[start + kq]), fix_word((unsigned)words[start + kq]));
        B.fx = r0.x;
        B.fy = r0.y;
        uint2* out = rec + B.base + lane;
        for (int t = 0; t < B.trips; ++t) {
            const int e = 4 * t + kq;
            uint2 r = nothing;
            if (e < cnt) r = make_uint2(__float_as_uint(vals[start + e]), fix_word((unsigned)words[start + e]));
            out[t * 64] = r;
        }
    };
    static_assert(kMaxRowBlocks == 2, "one call per row block");
    one_block(std::integral_constant<int, 0>{});
    one_block(std::integral_constant<int, 1>{});
    if constexpr (NB == 3) one_block(std::integral_constant<int, 2>{});
    return fits_slice;
}

// ---- cluster variant: one graph on K workgroups (one CU each) --------------------------------------------------
// A lone workgroup's layer is 4 096 MFMA cycles on the SIMD that holds four of an N = 200 graph's 13 tiles plus one row
// block's chain of LDS round trips; neither shrinks inside one CU, and the reference calls its agent with ONE graph.  In
// this variant every workgroup of a graph builds the whole LDS image, but transforms and aggregates only the row blocks
// it owns (block index mod K: each of its waves has at most one tile and one block per layer).  What it has to share is
// Z1: it writes its rows to a global slice and pulls the other K - 1 workgroups' rows from there into its own bufB as
// soon as they show up (cluster_pull_rows).  The workgroups of a graph get block indices that are equal modulo 8, i.e.
// the same XCD and the same L2 (round-robin dispatch; checked once through HW_REG_XCC_ID, fault bit otherwise), so that
// "visible" only means "has left the CU": s_waitcnt on the writer, L1-bypassing loads on the reader - an agent-scope
// release would write the L2 back (tools/micro/xchg.hip: 4 - 5 K cycles per exchange through a progress word this way,
// 14 K - 200 K with __threadfence()).  Spins are bounded: a workgroup that never arrives costs a fault bit, not the machine.
// (load and wait are ONE asm statement: the compiler does not know that the result of an inline-asm load is still in
// flight and would otherwise be free to copy the registers before the data has arrived)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load_l2_x4(const float* p0, const float* p1, const float* p2, const float* p3, float4 (&out)[4]) {
    f32x4_t v0, v1, v2, v3;
    asm volatile("global_load_dwordx4 %0, %4, off sc0 sc1\n\tglobal_load_dwordx4 %1, %5, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc0 sc1\n\tglobal_load_dwordx4 %3, %7, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
    out[0] = make_float4(v0[0], v0[1], v0[2], v0[3]);
    out[1] = make_float4(v1[0], v1[1], v1[2], v1[3]);
    out[2] = make_float4(v2[0], v2[1], v2[2], v2[3]);
    out[3] = make_float4(v3[0], v3[1], v3[2], v3[3]);
}
__device__ __forceinline__ float load_l2_scalar(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

constexpr unsigned kUnwritten = 0xffffffffu;  // see cluster_pull_rows
__device__ __forceinline__ float poll_l2_scalar(const float* p, int32_t* status) {
    for (int spins = 0;; ++spins) {
        const float v = load_l2_scalar(p);
        if (__float_as_uint(v) != kUnwritten) return v;
        if (spins > (1 << 19)) {
            if (status) atomicOr(status, DGCN_FAULT_CLUSTER);
            return v;
        }
    }
}

// every store of this workgroup has left the CU -> publish this launch's nonce
template <int BLOCK>
__device__ __forceinline__ void cluster_publish(unsigned long long* flags, int cw, unsigned long long nonce) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&flags[cw], nonce, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until all K workgroups have published it.  EQUALITY with a 64-bit value that no other launch of this process ever
// uses (fused_prepare): whatever the caller's workspace held before - an earlier launch's words, float data, records of an
// ordinary launch, a recycled allocation - reads as "not yet" with a 2^-64 exception per word, so the words need no
// clearing (round 2 compared "not older than this launch's epoch" modulo 2^32, which arbitrary bits satisfy half the time).
template <int BLOCK>
__device__ __forceinline__ void cluster_wait(unsigned long long* flags, int K, unsigned long long nonce, int32_t* status) {
    if ((int)threadIdx.x < K) {
        int spins = 0;
        while (__hip_atomic_load(&flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != nonce) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 21)) {  // ~0.3 s: somebody is not coming
                if (status) atomicOr(status, DGCN_FAULT_CLUSTER);
                break;
            }
        }
    }
    __syncthreads();
}
// A chunk of the exchange slice that has not been written yet holds four of these (a NaN no arithmetic produces); the
// reader pulls the DATA until no chunk shows it, so a hand-over is two L2 round trips - the writer's store, the
// reader's load - instead of three (store, progress word, load).  Three slices per graph, used in turn: after a workgroup
// has pulled layer l (=> every other workgroup has published l and is therefore done reading l - 1) it marks its own rows
// in the slice of layer l + 2 (= that of l - 1) unwritten again; s_waitcnt + the barrier after the gather phase put those
// marks into L2 before its next layer's rows leave, and nobody polls for l + 2 before having seen those.
// A workgroup's <= 8 tiles on its 8 waves: with one or two tiles a tile's four 16-column blocks go to four waves (8 MFMAs
// each instead of 32 in a row on one SIMD), with three or four to two waves, with more a wave has a tile to itself.  Every output element still sees the same
// eight MFMAs in the same order.
struct ClusterTile {
    int trow;      // row of lane & 15 in this wave's tile, -1 = none
    int ct0, nct;  // this wave's column blocks [ct0, ct0 + nct) of the tile's four (Z0: 0, 1; Z1: 2, 3)
};
template <int BLOCK>
__device__ __forceinline__ void cluster_tile_init(ClusterTile& ct, int ng, const unsigned short* perm, int K, int cw) {
    static_assert(BLOCK == 512, "eight waves");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blocks = (ng + 15) >> 4;
    const int owned = blocks > cw ? (blocks - cw + K - 1) / K : 0;
    const int wpt = owned <= 2 ? 4 : owned <= 4 ? 2 : 1;  // waves per tile
    const int blk = (wave / wpt) * K + cw;
    const int tslot = blk * 16 + (lane & 15);
    ct.trow = (blk < blocks && tslot < ng) ? (int)perm[tslot] : -1;
    ct.nct = 4 / wpt;
    ct.ct0 = (wave % wpt) * ct.nct;
}

template <int BLOCK>
__device__ __forceinline__ void cluster_mark_unwritten(const ClusterTile& t, float* slice) {
    const int kq = (threadIdx.x & 63) >> 4;
    const float u = __uint_as_float(kUnwritten);
    if (t.trow < 0 || t.ct0 != 0) return;  // one wave per tile
    float* row = slice + t.trow * kHid;
    *reinterpret_cast<float4*>(row + (kq << 2)) = make_float4(u, u, u, u);
    *reinterpret_cast<float4*>(row + ((kq + 4) << 2)) = make_float4(u, u, u, u);
}

// the other workgroups' Z1 rows: global slice (same swizzled 128-byte rows as bufB) -> bufB, as soon as they are there
template <int BLOCK>
__device__ __forceinline__ void cluster_pull_rows(const float* slice, float* bufB, const unsigned short* perm, int ng, int K, int cw,
                                                  int32_t* status) {
    const int total = ng * 8;  // 16-byte chunks, in perm order: position p belongs to block p / 16
    for (int base = threadIdx.x; base < total; base += 4 * BLOCK) {
        float4 v[4];
        int dst[4];
        const float* src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * BLOCK;
            dst[u] = -1;
            src[u] = slice;  // lanes without work read the slice's first bytes and drop them
            if (i < total && ((i >> 7) % K) != cw) {  // i >> 3 = position, >> 4 more = block
                const int row = perm[i >> 3];
                dst[u] = row * kHid + ((i & 7) << 2);
                src[u] = slice + dst[u];
            }
        }
        int spins = 0;
        while (true) {
            load_l2_x4(src[0], src[1], src[2], src[3], v);
            bool ready = true;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dst[u] >= 0 && (__float_as_uint(v[u].x) == kUnwritten || __float_as_uint(v[u].y) == kUnwritten ||
                                    __float_as_uint(v[u].z) == kUnwritten || __float_as_uint(v[u].w) == kUnwritten))
                    ready = false;
            if (ready) break;
            if (++spins > (1 << 19)) {  // ~0.3 s: somebody is not coming (or a row really holds that NaN)
                if (status) atomicOr(status, DGCN_FAULT_CLUSTER);
                break;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dst[u] >= 0) *reinterpret_cast<float4*>(bufB + dst[u]) = v[u];
    }
}

// hidden transform of the tiles this workgroup owns (rows in perm order); Z1 goes to bufB AND to the graph's exchange
// slice.  Z0 replaces the tile's rows of H in bufA, and several waves read those: barrier between the reads and the writes.
template <int BLOCK>
__device__ __forceinline__ void hidden_transform_owned(const float (&b)[8][4], const ClusterTile& t, float* bufA, float* bufB,
                                                       float* slice) {
    const int lane = threadIdx.x & 63;
    const int kq = lane >> 4;
    const int row = t.trow >= 0 ? t.trow : 0;  // lanes past the graph's end feed row 0 and write nothing
    float av[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) av[s] = bufA[row * kHid + (((s ^ (row & 7)) << 2) | kq)];
    __syncthreads();
    if (!__any(t.trow >= 0)) return;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        if (ct < t.ct0 || ct >= t.ct0 + t.nct) continue;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[s][ct], av[s], acc, 0, 0, 0);
        if (t.trow >= 0) {
            const int chunk = (ct & 1) * 4 + kq;
            const float4 o = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if (ct < 2) *reinterpret_cast<float4*>(bufA + row * kHid + ((chunk ^ (row & 7)) << 2)) = o;
            else {
                const int off = row * kHid + ((chunk ^ keyB(row)) << 2);
                *reinterpret_cast<float4*>(bufB + off) = o;
                *reinterpret_cast<float4*>(slice + off) = o;
            }
        }
    }
}

// ... and with the chains in double (layer index 1; fragments from load_bfrag(.., f64map = true), see hidden_transform_f64)
template <int BLOCK>
__device__ __forceinline__ void hidden_transform_owned_f64(const float (&b)[8][4], const ClusterTile& t, float* bufA, float* bufB,
                                                           float* slice) {
    const int lane = threadIdx.x & 63;
    const int kq = lane >> 4;
    const int row = t.trow >= 0 ? t.trow : 0;
    float av[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) av[s] = bufA[row * kHid + (((s ^ (row & 7)) << 2) | kq)];
    __syncthreads();
    if (!__any(t.trow >= 0)) return;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        if (ct < t.ct0 || ct >= t.ct0 + t.nct) continue;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)b[s][ct], (double)av[s], acc, 0, 0, 0);
        if (t.trow >= 0) {
            const int chunk = (ct & 1) * 4 + kq;
            const float4 o = make_float4((float)acc[0], (float)acc[1], (float)acc[2], (float)acc[3]);
            if (ct < 2) *reinterpret_cast<float4*>(bufA + row * kHid + ((chunk ^ (row & 7)) << 2)) = o;
            else {
                const int off = row * kHid + ((chunk ^ keyB(row)) << 2);
                *reinterpret_cast<float4*>(bufB + off) = o;
                *reinterpret_cast<float4*>(slice + off) = o;
            }
        }
    }
}

// (Tried: one progress word per tile instead of per workgroup - a wave publishes its tile as soon as its stores have left
// the CU, the readers' waves wait for and pull two foreign tiles at a time, no workgroup barrier around the hand-over:
// 120 - 128 us against 117 - 123 us for one to eight N = 200 graphs.  Three dependent L2 round trips - store, poll, pull -
// are what the hand-over costs either way.)

// A row's four-feature accumulator: float32 fmaf chain, or (F64: layer index 0) one fma chain in double that is rounded
// once after "+ Z0 (+ bias)".  In both, an entry of value -0.0f on the zero row leaves the chain as it is.
template <bool F64> struct RowAcc;
template <> struct RowAcc<false> {
    float4 v;
    __device__ __forceinline__ void clear() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void add(float a, float4 z) { v = fma4(a, z, v); }
    template <bool BIAS> __device__ __forceinline__ float4 finish(float4 y, float4 b) const {
        float4 o = make_float4(y.x + v.x, y.y + v.y, y.z + v.z, y.w + v.w);
        if constexpr (BIAS) { o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w; }
        return o;
    }
};
template <> struct RowAcc<true> {
    double x, y, z, w;
    __device__ __forceinline__ void clear() { x = y = z = w = 0.0; }
    __device__ __forceinline__ void add(float a, float4 q) {
        const double ad = (double)a;
        x = fma(ad, (double)q.x, x); y = fma(ad, (double)q.y, y); z = fma(ad, (double)q.z, z); w = fma(ad, (double)q.w, w);
    }
    __device__ __forceinline__ void add_d(double ad, const double (&q)[4]) {
        x = fma(ad, q[0], x); y = fma(ad, q[1], y); z = fma(ad, q[2], z); w = fma(ad, q[3], w);
    }
    template <bool BIAS> __device__ __forceinline__ float4 finish(float4 y0, float4 b) const {
        double ox = (double)y0.x + x, oy = (double)y0.y + y, oz = (double)y0.z + z, ow = (double)y0.w + w;
        if constexpr (BIAS) { ox += (double)b.x; oy += (double)b.y; oz += (double)b.z; ow += (double)b.w; }
        return make_float4((float)ox, (float)oy, (float)oz, (float)ow);
    }
};

// ---- aggregation at width 32: 4 lanes x 2 float4 per row, 16 rows per wave pass, rows in `perm` order (descending
// entry count; only the processing order changes, never the arithmetic).  Per row: sequential fmaf chain over the
// row's entries, slot by slot.  (Round 2 started with 8 lanes x float4 per row and 8 rows per pass: twice the
// prologue / tail / epilogue instructions per row for the same gathers.)  A lane (slot s = lane / 4, kq = lane % 4) owns chunks kq and
// kq + 4 of its row; slots with (s >> 1) & 1 set read the upper half first.  A ds_read_b128 is served in bank groups
// of 16 lanes = 4 row slots here ({0,3,5,6}, {1,2,4,7}, ... of MI355X_MICROARCH.md's lane sets): two of them read
// lower halves and two upper halves, so a group collides only where two rows that read the same half have the same
// parity.  The chunk swizzle inside a half (keyB) only serves the transform's ds_write_b128.
template <int BLOCK, int ACT, bool BIAS, bool F64 = false, int NB = 2>
__device__ __forceinline__ void aggregate_rows16(const float* bias_ptr, float* bufA, const RowBlocks& rb,
                                                 const uint2* rec, const unsigned* rinfo_lds, unsigned long long* st, bool const_rows = false) {
    (void)st;
    (void)rinfo_lds;
#ifdef DGCN_DIAG
// (accumulated in registers, written once at the end of the call: a global read-modify-write per stamp put a round trip into
// every phase it measured)
#define BSTAMP(i) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); bacc[i] += _t - bt; bt = _t; } while (0)
    unsigned long long bt = 0;
    unsigned long long bacc[6] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#else
#define BSTAMP(i) do { } while (0)
#endif
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int s = lane >> 2, kq = lane & 3;
    (void)wave;
    const int cfirst = kq | (((s >> 1) & 1) << 2), csecond = cfirst ^ 4;  // logical chunks of this lane
    const unsigned cA = (unsigned)cfirst << 4, cB = (unsigned)csecond << 4;
    constexpr int kWaves = BLOCK / 64;
    // (a template argument, like the activation: the reference's GCN_DQN layers have no bias, and eight registers of
    // -0.0f carried through the entry walk are eight too many on this kernel's register budget)
    float4 biasA = make_float4(0.f, 0.f, 0.f, 0.f), biasB = biasA;
    if constexpr (BIAS) {
        biasA = *reinterpret_cast<const float4*>(bias_ptr + 4 * cfirst);
        biasB = *reinterpret_cast<const float4*>(bias_ptr + 4 * csecond);
    }
    (void)kWaves;
    // (one call per block, not an unrolled loop: see row_blocks_init)
    auto one_block = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const RowBlock& B = rb.template at<k>();
        const int trips = B.trips;  // (wave-uniform: a scalar register)
        if (trips == 0) return;
        if (B.v < 0) return;        // (slots past the graph's last row: their lanes sit the block out)
#ifdef DGCN_DIAG
        bt = __builtin_amdgcn_s_memtime();
#endif
        int v = B.v;
        asm volatile("" : "+v"(v));  // (opaque: what derives from the row - swizzle keys, addresses - is formed here, per layer,
                                     // instead of being hoisted out of the layer loop into registers the kernel does not have)
        RowAcc<F64> accA, accB;
        accA.clear();
        accB.clear();
        // (the row's own Z0 chunks are requested ahead of the gathers: while four row blocks' worth of state lived through
        // the layer loop these 8 registers spilled elsewhere and cost 2 %; with two it is 0.5 % the other way)
        float4* ownA = reinterpret_cast<float4*>(bufA + v * kHid + ((cfirst ^ (v & 7)) << 2));
        float4* ownB = reinterpret_cast<float4*>(bufA + v * kHid + ((csecond ^ (v & 7)) << 2));
        const float4 yA = *ownA, yB = *ownB;  // the row's own Z0 chunks, requested ahead of the gathers
        BSTAMP(0);
        // Entry metadata comes from GLOBAL memory, not from the LDS: the gather phase is paced by the LDS instruction
        // stream (an LDS round trip takes ~450 cycles there: 16 waves x 8 ds_read_b128 queued), and the two metadata
        // reads per trip were a quarter of it.  A lane loads ONE 8-byte record {value, word} per trip - entry 4 t + kq of
        // its row: record 64 t + lane of the block, 512 consecutive bytes per wave (row_blocks_init) - one trip ahead;
        // the four lanes of a row then hand their entries round with quad-broadcast DPP moves (the row's lanes are a quad).
        // (Tried and dropped: issuing the 8 gathers of trip t + 1 before the 16 packed FMAs of trip t - uniform trip count
        // from the block's first row, ping-pong buffers, no spills: 211.8 us against 205.2 us per C3 launch.)
        const char* bp = reinterpret_cast<const char*>(rec + B.base);  // wave-uniform: scalar registers
        const unsigned voff = (unsigned)lane * 8u;
        uint2 cur = make_uint2(B.fx, B.fy);
        asm volatile("" : "+v"(cur.x), "+v"(cur.y));  // (opaque: or the compiler merges this with the loop's load into ONE load through
                                                      // a pointer that starts at the kernel's stack - a flat load, used at once, per trip)
#define DGCN_QB(x, e) __builtin_amdgcn_update_dpp(0, (int)(x), (e) * 0x55, 0xf, 0xf, true)
#define DGCN_TRIP(R, NE)                                                                                         \
        {                                                                                                      \
            float4 zA[NE], zB[NE];                                                                             \
            float av[NE];                                                                                      \
            _Pragma("unroll") for (int e = 0; e < NE; ++e) {                                                   \
                const unsigned w = (unsigned)(e == 0 ? DGCN_QB(R.y, 0) : e == 1 ? DGCN_QB(R.y, 1)          \
                                              : e == 2 ? DGCN_QB(R.y, 2) : DGCN_QB(R.y, 3));               \
                av[e] = __int_as_float(e == 0 ? DGCN_QB(R.x, 0) : e == 1 ? DGCN_QB(R.x, 1)                 \
                                       : e == 2 ? DGCN_QB(R.x, 2) : DGCN_QB(R.x, 3));                      \
                zA[e] = lds_chunk(w ^ cA);                                                                     \
                zB[e] = lds_chunk(w ^ cB);                                                                     \
            }                                                                                                  \
            _Pragma("unroll") for (int e = 0; e < NE; ++e) {                                                   \
                accA.add(av[e], zA[e]);                                                                        \
                accB.add(av[e], zB[e]);                                                                        \
            }                                                                                                  \
        }
        // (chains in double: two entries' gathers in flight instead of four - with 16 accumulator registers and the
        // conversions on top, four would push the row-block state into scratch memory; once per launch, the pace does not matter)
#define DGCN_TRIP_PAIR(R, E0, NE)                                                                                \
        {                                                                                                      \
            _Pragma("unroll") for (int e = (E0); e < (E0) + 2 && e < (NE); ++e) {                              \
                const unsigned w = (unsigned)(e == 0 ? DGCN_QB(R.y, 0) : e == 1 ? DGCN_QB(R.y, 1)          \
                                              : e == 2 ? DGCN_QB(R.y, 2) : DGCN_QB(R.y, 3));               \
                const float a1 = __int_as_float(e == 0 ? DGCN_QB(R.x, 0) : e == 1 ? DGCN_QB(R.x, 1)        \
                                                : e == 2 ? DGCN_QB(R.x, 2) : DGCN_QB(R.x, 3));             \
                accA.add(a1, lds_chunk(w ^ cA));                                                               \
                accB.add(a1, lds_chunk(w ^ cB));                                                               \
            }                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
        }
        bool walked = false;
        if constexpr (F64) {
            if (const_rows) {
                // Constant input features (X == NULL: every reference script's case): all rows of Z1 are the SAME 32 numbers,
                // so the chain fma(val_j, Z1[u_j][c], acc) needs the entries' values only - no gathers.  Same chain, same bits.
                // (The neutral records do not serve here: -0.0 times a NEGATIVE feature is +0.0, and -0.0 + +0.0 is +0.0 - so
                // the entries past the row's end are skipped by count.  A quad = one row: the test is uniform in it.)
                const float4 fA = lds_chunk(cA), fB = lds_chunk(cB);  // row 0 of bufB (keyB(0) = 0: chunk c sits at byte 16 c)
                const double dA[4] = {(double)fA.x, (double)fA.y, (double)fA.z, (double)fA.w};
                const double dB[4] = {(double)fB.x, (double)fB.y, (double)fB.z, (double)fB.w};
                const int cnt = (int)(rinfo_lds[v] >> 16);
                for (int t = 0; t < trips; ++t) {
                    const uint2 nxt = *reinterpret_cast<const uint2*>(bp + (t + 1) * 512 + voff);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const double ad = (double)__int_as_float(e == 0 ? DGCN_QB(cur.x, 0) : e == 1 ? DGCN_QB(cur.x, 1)
                                                                 : e == 2 ? DGCN_QB(cur.x, 2) : DGCN_QB(cur.x, 3));
                        if (4 * t + e < cnt) {
                            accA.add_d(ad, dA);
                            accB.add_d(ad, dB);
                        }
                    }
                    cur = nxt;
                }
                walked = true;
            }
        }
        if (!walked) {
            if constexpr (F64) {
                for (int t = 0; t < trips; ++t) {
                    const uint2 nxt = *reinterpret_cast<const uint2*>(bp + (t + 1) * 512 + voff);
                    DGCN_TRIP_PAIR(cur, 0, 4) DGCN_TRIP_PAIR(cur, 2, 4)
                    cur = nxt;
                }
            } else {
                // two trips per turn, so that the record loaded ahead is used where it landed (no register moves)
                uint2 nxt;
                int t = 0;
                for (; t + 1 < trips; t += 2) {
                    nxt = *reinterpret_cast<const uint2*>(bp + (t + 1) * 512 + voff);
                    DGCN_TRIP(cur, 4)
                    cur = *reinterpret_cast<const uint2*>(bp + (t + 2) * 512 + voff);  // (past the last trip: the next block's first record, or slack)
                    DGCN_TRIP(nxt, 4)
                }
                if (trips & 1) DGCN_TRIP(cur, 4)
            }
        }
        BSTAMP(1);
#undef DGCN_TRIP_PAIR
#undef DGCN_TRIP
#undef DGCN_QB
        BSTAMP(2);
        float4 oA = accA.template finish<BIAS>(yA, biasA);
        float4 oB = accB.template finish<BIAS>(yB, biasB);
        oA.x = apply_act(oA.x, ACT); oA.y = apply_act(oA.y, ACT); oA.z = apply_act(oA.z, ACT); oA.w = apply_act(oA.w, ACT);
        oB.x = apply_act(oB.x, ACT); oB.y = apply_act(oB.y, ACT); oB.z = apply_act(oB.z, ACT); oB.w = apply_act(oB.w, ACT);
        *ownA = oA;
        *ownB = oB;
        BSTAMP(3);
#ifdef DGCN_DIAG
        bacc[4] += 1;
        bacc[5] += (unsigned long long)trips;
#endif
    };
    one_block(std::integral_constant<int, 0>{});
    one_block(std::integral_constant<int, 1>{});
    if constexpr (NB == 3) one_block(std::integral_constant<int, 2>{});
#ifdef DGCN_DIAG
    if (st && threadIdx.x == (BLOCK == 1024 ? 0 : BLOCK - 64)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) st[i] += bacc[i];
    }
#endif
#undef BSTAMP
}

// ---- cluster variant of the aggregation: 8 lanes x float4 per row, 8 rows per wave, 8 entries per trip.
// A wave's walk over a row is a chain of ~500-cycle trips whatever the load on the LDS (measured: 553 cycles per 4-entry
// trip with four waves on the CU, 504 with sixteen), so the phase is as long as the longest row block's chain.  With the
// CU to itself and 256 VGPRs a workgroup spreads its <= 4 tiles over all 8 waves (half a tile each), takes 8 entries per
// trip, and keeps the rows' records in registers for the whole layer loop (no L2 round trip inside the phase).  Same
// per-row arithmetic: sequential fmaf chain over the entries, slot by slot.
constexpr int kRecCache = 12;  // records per lane kept in registers: the first 48 entries of a row
struct ClusterRowSet {
    int v;                  // row of this lane's slot (lane / 8), -1 = none
    unsigned ri;            // its rinfo
    uint2 recs[kRecCache];  // entry 4i + (lane & 3) of the row
};
// A workgroup's <= 8 tiles on its 8 waves: wave pair p takes the rows of the workgroup's tiles p and p + 4 (lower half on
// the even wave, upper half on the odd one).  (Two named sets, not an array: see RowBlocks.)
struct ClusterRows {
    ClusterRowSet s0, s1;
};

template <int BLOCK, int SET>
__device__ __forceinline__ void cluster_row_set_init(ClusterRowSet& cr, int ng, const unsigned* rinfo, const unsigned short* perm,
                                                     const float* vals, const unsigned short* words, int K, int cw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blk = ((wave >> 1) + 4 * SET) * K + cw;          // tile `wave / 2` (+ 4) of this workgroup
    const int slot = blk * 16 + (wave & 1) * 8 + (lane >> 3);  // its lower / upper half
    const int blocks = (ng + 15) >> 4;
    cr.v = -1;
    cr.ri = 0u;
#pragma unroll
    for (int i = 0; i < kRecCache; ++i) cr.recs[i] = make_uint2(0u, 0u);
    if (blk < blocks && slot < ng) {
        cr.v = perm[slot];
        cr.ri = rinfo[cr.v];
        const int rs = (int)(cr.ri & 0xffff), cnt = (int)(cr.ri >> 16);
#pragma unroll
        for (int i = 0; i < kRecCache; ++i)
            if (4 * i < cnt) {  // (straight from the LDS image: same bits as the global records, no L2 round trip)
                const int e = rs + 4 * i + (lane & 3);
                cr.recs[i] = make_uint2(__float_as_uint(vals[e]), (unsigned)words[e]);
            }
    }
}
template <int BLOCK>
__device__ __forceinline__ void cluster_rows_init(ClusterRows& cr, int ng, const unsigned* rinfo, const unsigned short* perm,
                                                  const float* vals, const unsigned short* words, int K, int cw) {
    cluster_row_set_init<BLOCK, 0>(cr.s0, ng, rinfo, perm, vals, words, K, cw);
    cluster_row_set_init<BLOCK, 1>(cr.s1, ng, rinfo, perm, vals, words, K, cw);
}

template <int BLOCK, int ACT, bool F64 = false>
__device__ __forceinline__ void aggregate_rows8c(const float* bias_ptr, float* bufA, const ClusterRowSet& cr, const uint2* rec,
                                                 unsigned zrow) {
    const int lane = threadIdx.x & 63;
    const int q = lane & 7, kq = lane & 3;
    const unsigned cq = (unsigned)q << 4;
    float4 bias = make_float4(-0.f, -0.f, -0.f, -0.f);  // x + (-0.0f) == x for every x (float and double alike)
    if (bias_ptr) bias = *reinterpret_cast<const float4*>(bias_ptr + 4 * q);
    if (!__any(cr.v >= 0) || cr.v < 0) return;
    const int v = cr.v;
    const int rs = (int)(cr.ri & 0xffff), re = rs + (int)(cr.ri >> 16);
    RowAcc<F64> acc;
    acc.clear();
    float4* own = reinterpret_cast<float4*>(bufA + v * kHid + ((q ^ (v & 7)) << 2));
    const float4 y = *own;  // the row's Z0 chunk, requested ahead of the gathers (registers are plentiful here)
    int j = rs;
#define DGCN_QB(x, e) __builtin_amdgcn_update_dpp(0, (int)(x), (e) * 0x55, 0xf, 0xf, true)
#define DGCN_PICK(x, e) ((e) == 0 ? DGCN_QB(x, 0) : (e) == 1 ? DGCN_QB(x, 1) : (e) == 2 ? DGCN_QB(x, 2) : DGCN_QB(x, 3))
    // one trip: entries 0..3 from record r0 (lane kq of the quad holds entry kq), 4..7 from r1
#define DGCN_TRIP8C(r0, r1)                                                                          \
    {                                                                                                \
        float4 z[8];                                                                                 \
        float av[8];                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                              \
            const unsigned w = (unsigned)(e < 4 ? DGCN_PICK(r0.y, e & 3) : DGCN_PICK(r1.y, e & 3));   \
            av[e] = __int_as_float(e < 4 ? DGCN_PICK(r0.x, e & 3) : DGCN_PICK(r1.x, e & 3));          \
            z[e] = lds_chunk(w ^ cq);                                                                \
        }                                                                                            \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) acc.add(av[e], z[e]);                          \
    }
    const uint2 nothing = make_uint2(0x80000000u, zrow);  // value -0.0f on the zero row: fmaf leaves acc as it is
#pragma unroll
    for (int i = 0; i < kRecCache; i += 2) {
        if (!__any(j < re)) break;
        if (j < re) {
            uint2 r0 = cr.recs[i], r1 = cr.recs[i + 1];
            if (j + kq >= re) r0 = nothing;       // past the row's end: neutralised in the lane that owns the entry
            if (j + 4 + kq >= re) r1 = nothing;
            DGCN_TRIP8C(r0, r1)
            j += 8;
        }
    }
    for (; j < re; j += 8) {  // rows of more than 48 entries: the rest from global memory
        uint2 r0 = rec[j + kq], r1 = rec[j + 4 + kq];
        if (j + kq >= re) r0 = nothing;
        if (j + 4 + kq >= re) r1 = nothing;
        DGCN_TRIP8C(r0, r1)
    }
#undef DGCN_TRIP8C
#undef DGCN_PICK
#undef DGCN_QB
    float4 o = acc.template finish<true>(y, bias);
    o.x = apply_act(o.x, ACT); o.y = apply_act(o.y, ACT); o.z = apply_act(o.z, ACT); o.w = apply_act(o.w, ACT);
    *own = o;
}

template <int BLOCK>
__device__ __forceinline__ void cluster_aggregate_set(const FusedLayer& L, float* bufA, const ClusterRowSet& cr, const uint2* rec,
                                                      unsigned zrow, bool precise) {
    const float* bias = L.bias;
    const int act = L.act;
    if (precise) {  // layer index 0: chains in double
        if (act == DGCN_ACT_RELU) aggregate_rows8c<BLOCK, DGCN_ACT_RELU, true>(bias, bufA, cr, rec, zrow);
        else if (act == DGCN_ACT_LEAKY_RELU) aggregate_rows8c<BLOCK, DGCN_ACT_LEAKY_RELU, true>(bias, bufA, cr, rec, zrow);
        else aggregate_rows8c<BLOCK, DGCN_ACT_LINEAR, true>(bias, bufA, cr, rec, zrow);
        return;
    }
    if (act == DGCN_ACT_RELU) aggregate_rows8c<BLOCK, DGCN_ACT_RELU>(bias, bufA, cr, rec, zrow);
    else if (act == DGCN_ACT_LEAKY_RELU) aggregate_rows8c<BLOCK, DGCN_ACT_LEAKY_RELU>(bias, bufA, cr, rec, zrow);
    else aggregate_rows8c<BLOCK, DGCN_ACT_LINEAR>(bias, bufA, cr, rec, zrow);
}
template <int BLOCK>
__device__ __forceinline__ void cluster_aggregate(const FusedLayer& L, float* bufA, const ClusterRows& cr, const uint2* rec, unsigned zrow,
                                                  bool precise) {
    cluster_aggregate_set<BLOCK>(L, bufA, cr.s0, rec, zrow, precise);
    cluster_aggregate_set<BLOCK>(L, bufA, cr.s1, rec, zrow, precise);  // (tiles 5 .. 8 of a workgroup: nothing to do for most graphs)
}

// The activation is a template argument of the row loop: one uniform branch per layer instead of four per row block.
template <int BLOCK, int NB = 2>
__device__ __forceinline__ void hidden_aggregate(const FusedLayer& L, float* bufA, const RowBlocks& rb, const uint2* rec,
                                                 const unsigned* rinfo, bool precise, unsigned long long* st = nullptr, bool const_rows = false) {
    const float* bias = L.bias;
    const int act = L.act;
    if (precise) {  // layer index 0: chains in double (once per launch: not worth twelve instantiations, the bias is a runtime test there)
        if (bias) {
            if (act == DGCN_ACT_RELU) aggregate_rows16<BLOCK, DGCN_ACT_RELU, true, true, NB>(bias, bufA, rb, rec, rinfo, nullptr, const_rows);
            else if (act == DGCN_ACT_LEAKY_RELU) aggregate_rows16<BLOCK, DGCN_ACT_LEAKY_RELU, true, true, NB>(bias, bufA, rb, rec, rinfo, nullptr, const_rows);
            else aggregate_rows16<BLOCK, DGCN_ACT_LINEAR, true, true, NB>(bias, bufA, rb, rec, rinfo, nullptr, const_rows);
        } else {
            if (act == DGCN_ACT_RELU) aggregate_rows16<BLOCK, DGCN_ACT_RELU, false, true, NB>(bias, bufA, rb, rec, rinfo, nullptr, const_rows);
            else if (act == DGCN_ACT_LEAKY_RELU) aggregate_rows16<BLOCK, DGCN_ACT_LEAKY_RELU, false, true, NB>(bias, bufA, rb, rec, rinfo, nullptr, const_rows);
            else aggregate_rows16<BLOCK, DGCN_ACT_LINEAR, false, true, NB>(bias, bufA, rb, rec, rinfo, nullptr, const_rows);
        }
        return;
    }
    if (bias) {
        if (act == DGCN_ACT_RELU) aggregate_rows16<BLOCK, DGCN_ACT_RELU, true, false, NB>(bias, bufA, rb, rec, rinfo, st);
        else if (act == DGCN_ACT_LEAKY_RELU) aggregate_rows16<BLOCK, DGCN_ACT_LEAKY_RELU, true, false, NB>(bias, bufA, rb, rec, rinfo, st);
        else aggregate_rows16<BLOCK, DGCN_ACT_LINEAR, true, false, NB>(bias, bufA, rb, rec, rinfo, st);
    } else {
        if (act == DGCN_ACT_RELU) aggregate_rows16<BLOCK, DGCN_ACT_RELU, false, false, NB>(bias, bufA, rb, rec, rinfo, st);
        else if (act == DGCN_ACT_LEAKY_RELU) aggregate_rows16<BLOCK, DGCN_ACT_LEAKY_RELU, false, false, NB>(bias, bufA, rb, rec, rinfo, st);
        else aggregate_rows16<BLOCK, DGCN_ACT_LINEAR, false, false, NB>(bias, bufA, rb, rec, rinfo, st);
    }
}

// Block-wide OR through dynamic LDS (hipcc's __syncthreads_or reserves 256 B of STATIC LDS, which would
// move the dynamic region - and with it bufB - off byte offset 0).  One barrier inside; the caller must
// pass another barrier before the next call re-uses the flag words.
template <int BLOCK>
__device__ __forceinline__ bool block_or(bool pred, unsigned* wflags) {
    const unsigned long long m = __ballot(pred);
    if ((threadIdx.x & 63) == 0) wflags[threadIdx.x >> 6] = m != 0ull;
    __syncthreads();
    unsigned any = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) any |= wflags[w];
    return any != 0;
}

// One wave turns the entry-count histogram into start offsets of a descending counting sort (lane i owns
// bins 9i..9i+8, suffix-scanned with shuffles).
__device__ __forceinline__ void hist_to_offsets(int* hist) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int loc[9], sum = 0;
#pragma unroll
        for (int i = 8; i >= 0; --i) { loc[i] = sum; sum += hist[lane * 9 + i]; }  // within-lane suffix (higher bins first)
        int above = sum;  // inclusive suffix over lanes >= lane, then made exclusive
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_down(above, off);
            if (lane + off < 64) above += t;
        }
        above -= sum;
#pragma unroll
        for (int i = 0; i < 9; ++i) hist[lane * 9 + i] = above + loc[i];
    }
}

// Greedy rounds over the rank keys in LDS (residual-graph variant).  A live vertex wins a round iff its
// rank is below every live neighbour's (`central`: iff it holds rank 0 - the global best); winners get
// mark[v] = 1, their live neighbours mark[u] = kill_mark (when non-zero), both leave the graph.
// Returns the number of rounds run.  The caller passes a barrier before and after.
template <int BLOCK>
__device__ __forceinline__ int greedy_rounds(unsigned short* key, uint8_t* mark, int kill_mark,
                                             const unsigned short* words, int rs, int re, int vv, int sub, int lpv,
                                             bool mine, unsigned* wflags, int max_rounds, bool central) {
    constexpr unsigned kDead = 0xFFFFu;
    int rounds = 0;
    while (true) {
        const unsigned mykey = mine ? (unsigned)key[vv] : kDead;
        const bool live = mykey != kDead;
        unsigned m = kDead;
        if (live && !central) {
            for (int j = rs + sub; j < re; j += lpv) {
                const int u = words[j] >> 7;
                const unsigned k = key[u];
                if (u != vv) m = min(m, k);
            }
        }
        for (int off = 1; off < lpv; off <<= 1) m = min(m, (unsigned)__shfl_xor((int)m, off));
        const bool won = live && (central ? mykey == 0u : mykey < m);
        if (!block_or<BLOCK>(live, wflags)) break;  // its barrier also orders every rank read before the kills below
        ++rounds;
        if (won) {
            for (int j = rs + sub; j < re; j += lpv) {
                const int u = words[j] >> 7;
                if (u != vv && key[u] != kDead) {
                    key[u] = (unsigned short)kDead;
                    if (kill_mark) mark[u] = (uint8_t)kill_mark;
                }
            }
            if (sub == 0) { key[vv] = (unsigned short)kDead; mark[vv] = 1; }
        }
        __syncthreads();
        if (central || (max_rounds > 0 && rounds >= max_rounds)) break;
    }
    return rounds;
}

// Block-wide sum of one double per thread in a fixed order (shuffle tree inside a wave, then the wave
// partials in wave order); result valid on every thread.  One barrier + one trailing barrier.
template <int BLOCK>
__device__ __forceinline__ double block_sum(double part, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int w = 1; w < BLOCK / 64; ++w) r += red[w];
    __syncthreads();
    return r;
}

// Ranks under (value desc, index asc) of the ng <= 512 vertices of an image (every one of them takes part):
//     rank[v] = #{w : val[w] > val[v]} + #{w < v : val[w] == val[v]}
// NK arrays at once (the rollout ranks priorities and weights).  First form: a lane or two per vertex, looping over every w -
// one float64 LDS read per PAIR of vertices, and the LDS pipe's time with it: 49 of a rollout step's 306 us at 500 vertices
// (profiles/r04_residual_step_phases.txt).  Here a lane keeps four vertices' values in registers and a wave meets each w of
// its share once, a read all its lanes share: a quarter-thousand reads per wave instead of sixty thousand, two VALU
// instructions per pair.  The waves split the w range; their partial counts meet in LDS counters.  Only `>` is counted: values
// that are all different give counts that add up to ng (ng - 1) / 2, equal values share a count and the sum falls short - then
// (and only then) every vertex adds the equal values below its index.  `cnt`: NK * cstride + NK words of LDS, cstride >= ng.
// Barriers inside; the caller passes one before (values written) and one after (ranks written).
// PREZEROED: the caller has cleared `cnt` and passed a barrier since (saves the one here).
template <int BLOCK, int NK, bool PREZEROED = false>
__device__ __forceinline__ void rank_blocked(int ng, const double* val0, const double* val1, unsigned* cnt, int cstride,
                                             unsigned short* out0, unsigned short* out1) {
    constexpr int W = BLOCK / 64, KV = 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), tv = threadIdx.x;
    if constexpr (!PREZEROED) {
        for (int i = threadIdx.x; i < NK * cstride + NK; i += BLOCK) cnt[i] = 0u;
        __syncthreads();
    }
    const int VB = (ng + 64 * KV - 1) / (64 * KV);  // blocks of 256 vertices: 1 or 2
    const int WS = W / VB;                          // waves per block: each takes a share of the w range
    const int vb = wave % VB, ws = wave / VB;
    if (ws < WS) {
        const int w_lo = ws * ng / WS, w_hi = (ws + 1) * ng / WS;
#pragma unroll
        for (int key = 0; key < NK; ++key) {
            const double* val = key ? val1 : val0;
            double pv[KV];
            int c[KV];
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                const int v = (vb * KV + k) * 64 + lane;
                pv[k] = v < ng ? val[v] : 1.0 / 0.0;
                c[k] = 0;
            }
            int w = w_lo;
            for (; w + 3 < w_hi; w += 4) {  // four reads in flight
                const double p0 = val[w], p1 = val[w + 1], p2 = val[w + 2], p3 = val[w + 3];
#pragma unroll
                for (int k = 0; k < KV; ++k) c[k] += (int)(p0 > pv[k]) + (int)(p1 > pv[k]) + (int)(p2 > pv[k]) + (int)(p3 > pv[k]);
            }
            for (; w < w_hi; ++w) {
                const double p0 = val[w];
#pragma unroll
                for (int k = 0; k < KV; ++k) c[k] += (int)(p0 > pv[k]);
            }
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                const int v = (vb * KV + k) * 64 + lane;
                if (v < ng && c[k]) atomicAdd(&cnt[key * cstride + v], (unsigned)c[k]);
            }
        }
    }
    __syncthreads();
    unsigned r[NK];
#pragma unroll
    for (int key = 0; key < NK; ++key) {
        r[key] = tv < ng ? cnt[key * cstride + tv] : 0u;
        unsigned sum = r[key];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += (unsigned)__shfl_xor((int)sum, off);
        if (lane == 0 && sum) atomicAdd(&cnt[NK * cstride + key], sum);
    }
    __syncthreads();
    const unsigned want = (unsigned)ng * (unsigned)(ng - 1) / 2u;
#pragma unroll
    for (int key = 0; key < NK; ++key) {
        if (cnt[NK * cstride + key] != want && tv < ng) {  // equal values somewhere (workgroup-uniform): index order among them
            const double* val = key ? val1 : val0;
            const double mine = val[tv];
            for (int w = 0; w < tv; ++w) r[key] += val[w] == mine;
        }
        if (tv < ng) (key ? out1 : out0)[tv] = (unsigned short)r[key];
    }
}

// Row order of the cluster variant: every workgroup of a graph must arrive at the SAME order (it decides who owns which
// rows), so the position of a row is its rank under (entry count desc, index asc), not the order atomics happened to
// take.  Keys = count << 16 | ~index in LDS (`key`: scratch for ng rounded up to 4 words), four per ds_read_b128, the
// vertices' scans split over 1, 2 or 4 lanes.  ng <= BLOCK.
template <int BLOCK>
__device__ __forceinline__ void rank_rows(int ng, const unsigned* rinfo, unsigned short* perm, unsigned short* ipos, unsigned* key) {
    const int ng4 = (ng + 3) & ~3;
    for (int v = threadIdx.x; v < ng4; v += BLOCK) key[v] = v < ng ? ((rinfo[v] >> 16) << 16) | (0xffffu - (unsigned)v) : 0u;
    __syncthreads();
    const int lp_log = (ng * 4 <= BLOCK) ? 2 : (ng * 2 <= BLOCK) ? 1 : 0;
    const int v = threadIdx.x >> lp_log, part = threadIdx.x & ((1 << lp_log) - 1);
    int pos = 0;
    if (v < ng) {
        const unsigned kv = key[v];
        for (int u0 = part * 4; u0 < ng4; u0 += 4 << lp_log) {
            const uint4 q = *reinterpret_cast<const uint4*>(key + u0);
            pos += (q.x > kv) + (q.y > kv) + (q.z > kv) + (q.w > kv);
        }
    }
    if (lp_log >= 1) pos += __shfl_xor(pos, 1);
    if (lp_log >= 2) pos += __shfl_xor(pos, 2);
    if (v < ng && part == 0) {
        perm[pos] = (unsigned short)v;
        ipos[v] = (unsigned short)pos;
    }
}

// this graph is done and its outputs have left the CU: count it, the last one tells the host (DoneHook, common.h)
__device__ __forceinline__ void signal_done(const FusedArgs& a) {
    if (!a.done_flag) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();  // outputs may still sit in this XCD's L2: write them back before anybody is told
        const unsigned prev = atomicAdd(a.done_count, 1u);
        if (prev + 1u == a.done_target) __hip_atomic_store(a.done_flag, (int32_t)a.done_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// COMPACT: the batch comes in the compact transfer form (common.h CompactHook): a kernel of its own, also by name - what a
// profile of the host-to-host path shows (solves of consecutive batches overlapping) does not mix into the resident launch's row.
// GW (round 6; plain solves of mixed batches such as the BA test mix, whose largest image - 146 KB at 300 vertices / 11 200
// entries - used to give EVERY graph of the launch a CU to itself, a kernel launch having one LDS size): the image keeps in LDS
// only what the layers touch - bufA, bufB, the row tables.  The entry VALUES live in global scratch (as in the GVALS variant),
// the 16-bit gather WORDS are built in the space bufA will occupy, written out to global scratch once the block-major records
// exist (the layers walk those, not the words) and read back into bufB's tail behind the last hidden layer, where the last
// layer and the greedy rounds find them as always.  300 vertices: 79.5 KB - TWO workgroups per CU, C3's regime; 512 threads
// then hold up to three row blocks per wave (384 vertices).  Never with MASKED or CLUSTER.
template <bool MASKED, bool GVALS, int BLOCK, bool CLUSTER = false, bool COMPACT = false, bool GW = false>
// (4 waves per SIMD = 128 VGPRs: what lets two 512-thread workgroups share a CU and a 1024-thread one launch at all)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(CLUSTER ? 2 : 4))) void k_fused(FusedArgs a) {
    static_assert(!GW || (GVALS && !MASKED && !CLUSTER && BLOCK == 512), "the words-in-global variant: plain solves, 512 threads, values in global scratch");
    constexpr int kNB = GW ? 3 : 2;  // row blocks per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    int g = blockIdx.x;
    if (!CLUSTER && a.order) g = a.order[g];
    int cw = 0;                                  // cluster variant: which of the graph's K workgroups this is
    const int K = CLUSTER ? a.cluster : 1;
    if constexpr (CLUSTER) {
        // graphs in groups of 8; the K workgroups of a graph have block indices equal modulo 8 (one XCD)
        const int per = 8 * K, grp = blockIdx.x / per, rem = blockIdx.x % per;
        cw = rem >> 3;
        g = grp * 8 + (rem & 7);
        if (g >= a.num_graphs) return;
    }
    unsigned long long* xfl = CLUSTER ? a.xflag + (size_t)g * 8 : nullptr;                                   // this graph's progress words
    int32_t* xcc_slots = CLUSTER ? reinterpret_cast<int32_t*>(a.xflag + (size_t)8 * ((a.num_graphs + 7) & ~7)) + (size_t)g * 8 : nullptr;  // ... XCC ids
    const bool stamp_wg = !CLUSTER || cw == 0;  // (DGCN_DIAG builds: phase clocks of the graph's first workgroup only)
    (void)stamp_wg;
    const int n0 = a.graph_ptr[g], n1 = a.graph_ptr[g + 1];
    int ng = n1 - n0;  // (residual-graph variant: becomes the number of REMAINING vertices once the image is built)
    // bufB (Z1) sits at LDS byte offset 0 - the kernel has no static LDS - so a gather address is the
    // metadata word xor-ed with the lane's chunk offset, with no base to add
    float* bufB = reinterpret_cast<float*>(lds_raw);
    float* bufA = bufB + (size_t)a.max_nodes * kHid;
    unsigned* rinfo = reinterpret_cast<unsigned*>(bufA + (size_t)a.max_nodes * kHid);
    unsigned* wflags = reinterpret_cast<unsigned*>(lds_raw + a.flags_off);  // [waves] block-wide OR scratch
    const unsigned zrow = (unsigned)a.flags_off + 128u;  // LDS byte address of 128 zero bytes
    if (threadIdx.x < 32) wflags[32 + threadIdx.x] = 0u;
    uint2* rec = a.grec + (size_t)g * a.rec_cap;  // (cluster variant: every workgroup of the graph writes the same records)
    float* lds_meta = reinterpret_cast<float*>(rinfo + ((a.max_nodes + 3) & ~3));
    float* vals = GVALS ? a.gvals + (size_t)g * a.meta_cap : lds_meta;
    // (GW: the words are built where bufA will be - behind P0's row starts -, see above)
    unsigned short* words = GW ? reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(bufA) + kGwWordsP0)
                               : reinterpret_cast<unsigned short*>(GVALS ? lds_meta : lds_meta + a.meta_cap);
    unsigned short* perm = GW ? reinterpret_cast<unsigned short*>(lds_meta) : words + a.meta_cap;
    unsigned short* ipos = perm + a.max_nodes;  // position of a vertex in `perm`
    if (ng <= 0) {
        if (threadIdx.x == 0 && a.do_lgs) {
            if (a.rounds) a.rounds[g] = 0;
            if (a.totals) a.totals[g] = 0.0;
        }
        if (!MASKED && cw == 0 && a.do_lgs) signal_done(a);
        return;
    }
    // Two workgroups share a CU and the hardware favours the older one: measured, the first-dispatched
    // workgroup ran a layer in ~9.5 us, the second in ~13.5 us, and the launch ends with the slower
    // half.  Waves of every second dispatch wave (observed placement: block b and b + #CUs share a CU;
    // speed only, never correctness) raise their issue priority to even the two out - for prio_second of every 8
    // layers: raised all the time the second workgroup wins by as much as it loses without (4 313 vs 4 812 and
    // 4 955 vs 4 182 hundred cycles per graph), and the launch ends with the slower one.
    const bool second = (blockIdx.x >> 8) & 1;
    if (a.prio_second && second) __builtin_amdgcn_s_setprio(1);
    // (Measured in round 5, DESIGN_HISTORY: starting the second co-resident workgroup half a layer late gains nothing; the two
    // workgroups taking turns in the aggregation phase through a word in global memory costs 23 %: profiles/r05_fused_turns.txt.)
    unsigned long long tclk = 0;
#ifdef DGCN_DIAG
    tclk = __builtin_amdgcn_s_memtime();
    unsigned long long stamp_acc[12] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
#endif
    (void)tclk;
#ifdef DGCN_DIAG
    if (a.stamps && stamp_wg && threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.stamps[(size_t)g * 64 + 12] = hw;
        a.stamps[(size_t)g * 64 + 13] = xcc;
        a.stamps[(size_t)g * 64 + 14] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const int e0 = COMPACT ? a.cedge[g] : a.row_ptr[n0], e1 = COMPACT ? a.cedge[g + 1] : a.row_ptr[n1];
    int fault = 0;

    // ------------------------------------------------------------ P0: the support matrix into LDS
    // Row v of L (diagonal first, then the adjacency row) occupies entry slots [S_v, S_v + c_v) with
    // S_v = (U_v + v + 1) & ~1 where U_v is the unpadded start: even, and never overlapping.
    // Two dependent global round trips only: (row_ptr, d^-1/2 table) then (col_idx [, values]), the
    // second one entry-parallel and coalesced with 4 loads in flight per thread; the owning row of
    // an entry is found by bisection over the row starts kept in LDS.
    double* dinv = reinterpret_cast<double*>(bufB);  // scratch until the first transform
    int* rowstart = reinterpret_cast<int*>(bufA);    // [ng + 1], scratch until the first transform
    int* hist = reinterpret_cast<int*>(dinv + a.max_nodes);  // [576] entry-count histogram (bufB: 128 B per row)
    float xfill = a.x_const;   // first-layer input when X is null
    bool was_alive = true;     // vertex threadIdx.x is part of the (residual) graph
    int vid = threadIdx.x;     // the vertex thread threadIdx.x stands for (residual-graph variant: see the renumbering below)
    if constexpr (MASKED) {
        // Residual graph: rows and entries of removed vertices (state != 0) are dropped while the image is
        // built, so everything after P0 runs unchanged on the induced subgraph - the reference re-slices
        // the SciPy matrix instead (mwis_gdpg_call.py:284-285).  Row-parallel (8 lanes per row) because
        // an entry's slot depends on how many earlier entries of its row survive.
        // Round 3: the remaining vertices are RENUMBERED 0 .. na - 1 (in index order, so every "lower index first" rule
        // holds) and everything after P0 - tiles, row blocks, ranks, rounds - sees a graph of na vertices: a rollout search
        // removes ~11 vertices per step, so over a search the transforms and aggregations shrink to half on average.
        // Thread c stands for remaining vertex `vid` = orig[c] wherever a per-vertex global array is touched.
        uint8_t* al = reinterpret_cast<uint8_t*>(rowstart + 520);              // [512] 1 = in the residual graph
        unsigned short* acount = reinterpret_cast<unsigned short*>(al + 512);  // [512] surviving neighbours (by new index)
        unsigned short* cidx = acount + 512;                                   // [512] old index -> new
        unsigned short* orig = cidx + 512;                                     // [512] new index -> old
        double* wred = reinterpret_cast<double*>(hist + 576);                  // per-wave partial maxima
        int* wcnt = reinterpret_cast<int*>(wred + BLOCK / 64);                 // per-wave counts of remaining vertices
        const int tv0 = threadIdx.x;
        was_alive = tv0 < ng && a.state[n0 + tv0] == 0;
        const double w0 = (tv0 < ng && a.weights) ? a.weights[n0 + tv0] : 1.0;
        if (tv0 < ng) al[tv0] = was_alive;
        for (int i = threadIdx.x; i < 576; i += BLOCK) hist[i] = 0;
        // what a removed vertex reports as its score (the others write theirs after the last layer)
        if (a.scores && tv0 < ng && !was_alive && !(a.options & DGCN_RESIDUAL_SCORES_GIVEN)) a.scores[n0 + tv0] = 0.f;
        // nothing left, or no positive weight left (np.sum(wts_nn) <= 0 -> break, mwis_gdpg_call.py:286)
        if (!block_or<BLOCK>(was_alive && w0 > 0.0, wflags)) {
            if (a.scores && tv0 < ng && !(a.options & DGCN_RESIDUAL_SCORES_GIVEN)) a.scores[n0 + tv0] = 0.f;
            if (threadIdx.x == 0) {
                if (a.rounds) a.rounds[g] = 0;
                if (a.totals) a.totals[g] = 0.0;
            }
            return;
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gw = lane >> 3, q = lane & 7;
        const unsigned long long alive_mask = __ballot(was_alive);
        {
            double mx = was_alive ? w0 : -1.0 / 0.0;
            if (a.feature_mode == 1)
                for (int off = 1; off < 64; off <<= 1) mx = fmax(mx, __shfl_xor(mx, off));
            if (lane == 0) { wred[wave] = mx; wcnt[wave] = __popcll(alive_mask); }
        }
        __syncthreads();
        double wmax = wred[0];
        int na = 0, before = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) {
            if (w > 0) wmax = fmax(wmax, wred[w]);
            if (w < wave) before += wcnt[w];
            na += wcnt[w];
        }
        if (was_alive) {
            const int c = before + __popcll(alive_mask & ((1ull << lane) - 1ull));
            cidx[tv0] = (unsigned short)c;
            orig[c] = (unsigned short)tv0;
        }
        __syncthreads();
        STAMP(a, g, 0, tclk);  // residual P0: states, weights, "anything left", renumbering
        const int ng_full = ng;
        if (a.tail_word && threadIdx.x == 0) atomicMax(a.tail_word, a.tail_tag | (unsigned long long)(unsigned)na);
        ng = na;
        vid = tv0 < na ? (int)orig[tv0] : 0;
        was_alive = tv0 < na;  // from here on: "thread tv0 stands for a vertex of the image"
        if (a.feature_mode == 1) {
            const double wv = (was_alive && a.weights) ? a.weights[n0 + vid] : 1.0;
            xfill = was_alive ? (float)(wv / (wmax + 1e-9)) : 0.f;
        }
        // Rows are walked 8 lanes to a row, BLOCK / 8 rows at a time, twice: to count the remaining neighbours and, once the
        // slots are known, to write the entries.  Every lane asks for the bounds of ALL its rows and then for the first 16
        // columns of each before it looks at any of them (one dependent pair of round trips instead of one per 128 rows and
        // pass: 18 -> 9 us of a 500-vertex step), and keeps them for the second walk.
        constexpr int kRowsPerIt = BLOCK / 8;
        constexpr int kIt = (kFusedMaxNodes + kRowsPerIt - 1) / kRowsPerIt;  // 4 (1024 threads) or 8
        int r_s[kIt], r_e[kIt], r_o[kIt], c0[kIt], c1[kIt];
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int v = wave * 8 + gw + it * kRowsPerIt;
            r_s[it] = r_e[it] = 0;
            r_o[it] = 0;
            if (v < ng) {
                r_o[it] = orig[v];
                r_s[it] = a.row_ptr[n0 + r_o[it]];
                r_e[it] = a.row_ptr[n0 + r_o[it] + 1];
            }
        }
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            c0[it] = r_s[it] + q < r_e[it] ? a.col_idx[r_s[it] + q] - n0 : -1;
            c1[it] = r_s[it] + 8 + q < r_e[it] ? a.col_idx[r_s[it] + 8 + q] - n0 : -1;
        }
        auto count_one = [&](int u, int vo, bool present) -> int {
            if (!present) return 0;
            if (u < 0 || u >= ng_full) { fault |= DGCN_FAULT_BAD_COLUMN; return 0; }
            if (u == vo) fault |= DGCN_FAULT_SELF_LOOP;
            return (int)al[u];
        };
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int v = wave * 8 + gw + it * kRowsPerIt;
            if (wave * 8 + it * kRowsPerIt >= ng) break;  // wave-uniform
            const int rs = r_s[it], re = r_e[it], vo = r_o[it];
            int cnt = count_one(c0[it], vo, rs + q < re) + count_one(c1[it], vo, rs + 8 + q < re);
            for (int j = rs + 16 + q; j < re; j += 8) cnt += count_one(a.col_idx[j] - n0, vo, true);
            cnt += __shfl_xor(cnt, 1);
            cnt += __shfl_xor(cnt, 2);
            cnt += __shfl_xor(cnt, 4);
            if (q == 0 && v < ng) acount[v] = (unsigned short)cnt;
        }
        __syncthreads();
        for (int v = threadIdx.x; v < ng; v += BLOCK) {
            const int vo = orig[v];
            const int rs = a.row_ptr[n0 + vo];
            const int start = ((rs - e0) + 2 * vo + 1) & ~1;  // the full row's slots stay reserved
            const int deg = acount[v];
            const int cnt = deg + 1;
            rinfo[v] = (unsigned)start | ((unsigned)cnt << 16);
            atomicAdd(&hist[min(cnt, 575)], 1);
            double d = 0.0;
            if (deg < a.table_len) d = a.dinv_table[deg]; else fault |= DGCN_FAULT_DEGREE_RANGE;
            dinv[v] = d;
            words[start] = enc_word(v);
            vals[start] = 1.0f;
        }
        __syncthreads();
        STAMP(a, g, 1, tclk);  // residual P0: row bounds, first columns, surviving neighbours, row slots, d^-1/2
        hist_to_offsets(hist);
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int v = wave * 8 + gw + it * kRowsPerIt;
            if (wave * 8 + it * kRowsPerIt >= ng) break;  // wave-uniform
            const bool act = v < ng;
            const int rs = r_s[it], re = r_e[it];
            const int start = act ? (int)(rinfo[v] & 0xffff) : 0;
            int base = 1;  // slot 0 of the row is the diagonal
            auto place = [&](int u) {  // (every lane of the wave comes through here together: the ballot)
                const bool keep = u >= 0 && u < ng_full && al[u];
                const unsigned bits = (unsigned)(__ballot(keep) >> (gw * 8)) & 0xffu;
                if (keep) {
                    const int slot = start + base + __popc(bits & ((1u << q) - 1u));
                    const int uc = cidx[u];
                    words[slot] = enc_word(uc);
                    vals[slot] = (float)(-(dinv[uc] * dinv[v]));
                }
                base += __popc(bits);
            };
            place(c0[it]);  // (-1 where the row has no such entry)
            if (__any(rs + 8 < re)) place(c1[it]);
            for (int j0 = rs + 16; __any(j0 < re); j0 += 8) {
                const int j = j0 + q;
                place(j < re ? a.col_idx[j] - n0 : -1);
            }
        }
        __syncthreads();
        if constexpr (CLUSTER) rank_rows<BLOCK>(ng, rinfo, perm, ipos, reinterpret_cast<unsigned*>(rowstart + 520 + 512 + 512));  // (behind al / acount / cidx / orig)
        else
        for (int v = threadIdx.x; v < ng; v += BLOCK) {
            const int c = min((int)(rinfo[v] >> 16), 575);
            const int pos = atomicAdd(&hist[c], 1);
            perm[pos] = (unsigned short)v;
            ipos[v] = (unsigned short)pos;
        }
    } else {
    for (int i = threadIdx.x; i < 576; i += BLOCK) hist[i] = 0;
    __syncthreads();
    const int extra = a.from_adj ? 1 : 0;            // the diagonal entry is synthesised from the adjacency
    // (compact batch: the row bounds are a block-wide exclusive scan of the degrees - one vertex per thread, ng <= BLOCK)
    int crs = 0, cdg = 0;
    if constexpr (COMPACT) {
        int* wsum = rowstart + 520;  // [BLOCK / 64] wave totals (scratch like rowstart itself)
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        cdg = (int)threadIdx.x < ng ? (int)a.cdeg[n0 + threadIdx.x] : 0;
        int inc = cdg;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) before += w < wave ? wsum[w] : 0;
        crs = e0 + before + inc - cdg;
    }
    for (int v = threadIdx.x; v < ng; v += BLOCK) {
        const int rs = COMPACT ? crs : a.row_ptr[n0 + v], re = COMPACT ? crs + cdg : a.row_ptr[n0 + v + 1];
        const int start = ((rs - e0) + v * extra + v + 1) & ~1;
        rinfo[v] = (unsigned)start | ((unsigned)(re - rs + extra) << 16);
        rowstart[v] = rs - e0;
        atomicAdd(&hist[min(re - rs + extra, 575)], 1);
        if (a.from_adj) {
            const int deg = re - rs;
            double d = 0.0;
            if (deg < a.table_len) d = a.dinv_table[deg]; else fault |= DGCN_FAULT_DEGREE_RANGE;
            dinv[v] = d;
            words[start] = enc_word(v);
            vals[start] = 1.0f;
        }
    }
    if (threadIdx.x == 0) rowstart[ng] = e1 - e0;
    __syncthreads();
    // Row order for the gather phase: counting sort by entry count, descending.  One wave turns the
    // histogram into start offsets (lane i owns bins 9i..9i+8, suffix-scanned with shuffles); the order
    // among equal counts is arbitrary - it only decides which rows share a lockstep pass.
    hist_to_offsets(hist);
    __syncthreads();
    STAMP(a, g, 0, tclk);  // P0a: row pointers, degree table
    {
        const int total = e1 - e0;
        constexpr int kP0 = 8;  // up to eight consecutive entries per thread: loads in flight, ONE bisection for their row
        const int chunk = min(kP0, max(1, (total + BLOCK - 1) / BLOCK));  // (small graphs: fewer per thread, all threads busy)
        for (int base = threadIdx.x * chunk; base < total; base += BLOCK * chunk) {
            int c[kP0];
            float gv[kP0];
#pragma unroll
            for (int i = 0; i < kP0; ++i) {
                const int j = base + i;
                c[i] = 0;
                gv[i] = 0.f;
                if (i < chunk && j < total) {
                    c[i] = COMPACT ? (int)a.ccol[e0 + j] + n0 : a.col_idx[e0 + j];
                    if (!a.from_adj) gv[i] = a.vals[e0 + j];
                }
            }
            int lo = 0, hi = ng;  // last row whose start is <= base (rows without entries are stepped over below)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (rowstart[mid] <= base) lo = mid; else hi = mid;
            }
            int v = lo;
            int vend = rowstart[v + 1];  // rowstart[ng] = total
            int vslot = (int)(rinfo[v] & 0xffff) + extra - rowstart[v];
            double dv = a.from_adj ? dinv[v] : 0.0;
#pragma unroll
            for (int i = 0; i < kP0; ++i) {
                const int j = base + i;
                if (i >= chunk || j >= total) break;
                if (j >= vend) {  // next row(s)
                    do { ++v; vend = rowstart[v + 1]; } while (j >= vend);
                    vslot = (int)(rinfo[v] & 0xffff) + extra - rowstart[v];
                    if (a.from_adj) dv = dinv[v];
                }
                int u = c[i] - n0;
                float val = gv[i];
                if (u < 0 || u >= ng) { fault |= DGCN_FAULT_BAD_COLUMN; u = 0; val = 0.f; }
                else if (a.from_adj) {
                    if (u == v) fault |= DGCN_FAULT_SELF_LOOP;
                    val = (float)(-(dinv[u] * dv));
                }
                const int slot = vslot + j;
                words[slot] = enc_word(u);
                vals[slot] = val;
            }
        }
        STAMP(a, g, 1, tclk);  // P0b: entries
        if constexpr (CLUSTER) {
            // every workgroup of the graph must arrive at the SAME row order (it decides who owns which rows): the
            // position of a row is its rank under (entry count desc, index asc), not the order atomics happened to take
            rank_rows<BLOCK>(ng, rinfo, perm, ipos, reinterpret_cast<unsigned*>(rowstart + 520));
        } else
        for (int v = threadIdx.x; v < ng; v += BLOCK) {
            const int c = min((int)(rinfo[v] >> 16), 575);
            const int pos = atomicAdd(&hist[c], 1);
            perm[pos] = (unsigned short)v;
            ipos[v] = (unsigned short)pos;
        }
    }
    }
    __syncthreads();  // scratch (bufA, bufB) is dead from here on
    // (Tried: a lone 1024-thread workgroup taking its entry metadata from the LDS instead of the global records - 136.0 vs
    // 132.5 us for one graph, 448 vs 455 us for the C4 share: no clear winner, one code path kept.)
    // (32-wide aggregations only: a one-layer model has none)
    const bool has_wide = !(MASKED && (a.options & DGCN_RESIDUAL_SCORES_GIVEN)) &&
                          (a.wide_passes > 1 || (a.num_layers > 1 && a.layers[0].cout == kHid));
    // (cluster variant: the first 4 * kRecCache entries of every row stay in registers; rows are in descending entry
    // order, so the global copy is needed only if the first one is longer than that)
    if constexpr (CLUSTER) {
        if (has_wide && (int)(rinfo[perm[0]] >> 16) > 4 * kRecCache) {
            // the support once more as 8-byte records in global memory, row-major (L2-resident: 19 layers re-read them)
            const unsigned rl = rinfo[ng - 1];
            const int used = min((int)(rl & 0xffff) + (int)(rl >> 16) + 8, a.meta_cap);
            for (int j = threadIdx.x; j < used; j += BLOCK)
                rec[j] = make_uint2(__float_as_uint(vals[j]), (unsigned)words[j]);
        }
        __syncthreads();
    }
    RowBlocks rb;
    ClusterRows cr;
    ClusterTile ctile;
    // (every wave writes the block-major records of its own row blocks: read back by the same lanes, no barrier)
    if constexpr (!CLUSTER) {
        if (!row_blocks_init<BLOCK, kNB>(rb, has_wide ? ng : 0, rinfo, perm, vals, words, rec, zrow, a.rec_cap, DIAG_ON(a, 4) != 0))
            fault |= DGCN_FAULT_DEGREE_RANGE;  // (a row with more entries than the graph has vertices: results invalid)
        if constexpr (GW) {
            // the words leave the LDS for the duration of the layers (two per 32-bit store; slots are even-aligned per row and
            // meta_cap is a multiple of 16)
            const unsigned rl = rinfo[ng - 1];
            const int used = min(((int)(rl & 0xffff) + (int)(rl >> 16) + 1) >> 1, a.meta_cap >> 1);
            unsigned* gw = reinterpret_cast<unsigned*>(a.gwords + (size_t)g * a.meta_cap);
            const unsigned* lw = reinterpret_cast<const unsigned*>(words);
            for (int j = threadIdx.x; j < used; j += BLOCK) gw[j] = lw[j];
            __syncthreads();  // every wave has read the words it needs: the first transform may overwrite them
        }
    }
    if constexpr (CLUSTER) {
        cluster_rows_init<BLOCK>(cr, has_wide ? ng : 0, rinfo, perm, vals, words, K, cw);
        cluster_tile_init<BLOCK>(ctile, has_wide ? ng : 0, perm, K, cw);
    }
    if constexpr (CLUSTER) {
        if (threadIdx.x == 0) {  // where this workgroup runs: compared after the first exchange
            unsigned id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
            __hip_atomic_store(&xcc_slots[cw], (int)(id & 15u) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // whatever an earlier launch left in the exchange slices: this workgroup's rows read "unwritten" from here on,
        // and the others learn through the progress word that they do (waited for in front of the first pull)
        for (int t = 0; t < 3; ++t) cluster_mark_unwritten<BLOCK>(ctile, a.xz + ((size_t)g * 3 + t) * a.max_nodes * kHid);
        if ((int)threadIdx.x < ng && ((ipos[threadIdx.x] >> 4) % K) == cw) {  // the last layer's scalars likewise
            float* xs0 = a.xs + (size_t)g * 2 * a.max_nodes;
            xs0[threadIdx.x] = __uint_as_float(kUnwritten);
            xs0[a.max_nodes + threadIdx.x] = __uint_as_float(kUnwritten);
        }
        cluster_publish<BLOCK>(xfl, cw, a.nonce);
    }
    STAMP(a, g, 2, tclk);  // P0c: row order

    // ------------------------------------------------------------ layers
    float score = 0.f;  // final output of vertex threadIdx.x (ng <= block, checked by the host)
    const bool scores_given = MASKED && (a.options & DGCN_RESIDUAL_SCORES_GIVEN);
    const int voff = vid - (int)threadIdx.x;  // 0 unless the residual-graph variant renumbered the vertices
    if (scores_given && (int)threadIdx.x < ng) score = a.scores[n0 + vid];
    float bfrag[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) bfrag[i][c] = 0.f;
    const bool wave_has_tile = (int)(threadIdx.x >> 6) < ((ng + 15) >> 4);  // hidden_transform: tile t on wave t % (BLOCK / 64)
    const int P = a.wide_passes > 1 ? a.wide_passes : 1;  // first-layer column blocks (1 = the ordinary case)
    int l_first = 0;
    if (P > 1 && !scores_given) {
        // Wide two-layer stack F -> c -> 1: block p of the first layer's columns goes through the 32-wide transform +
        // aggregation; its 32 features of H then feed terms 32p .. 32p+31 of the last layer's two chains (the k order
        // of the one-pass product), after which bufA is free for the next block.  The finished chains wait in
        // bufA[v] / bufB[v] for the last-layer code below.  (Kept apart from the deep-stack loop: nothing here is
        // live while that loop holds its MFMA operands.)
        const FusedLayer& LL = a.layers[P];
        const int v = threadIdx.x;
        double zc0 = 0.0, zc1 = 0.0;  // (the last layer is layer index 1: its chains run in double)
        for (int p = 0; p < P; ++p) {
            first_layer_transform<BLOCK>(a, a.layers[p], n0 + voff, ng, bufA, bufB, xfill);
            __syncthreads();
            hidden_aggregate<BLOCK, kNB>(a.layers[p], bufA, rb, rec, rinfo, true, nullptr, !a.X && !(MASKED && a.feature_mode == 1));  // (every block is layer index 0)
            __syncthreads();
            if (v < ng) {
#pragma unroll
                for (int c = 0; c < kHid / 4; ++c) {
                    const float4 h = *reinterpret_cast<const float4*>(bufA + v * kHid + ((c ^ (v & 7)) << 2));
                    const float hk[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        zc0 = fma((double)hk[i], (double)LL.W[(kHid * p + 4 * c + i) * 2 + 0], zc0);
                        zc1 = fma((double)hk[i], (double)LL.W[(kHid * p + 4 * c + i) * 2 + 1], zc1);
                    }
                }
            }
            __syncthreads();
        }
        if (v < ng) { bufA[v] = (float)zc0; bufB[v] = (float)zc1; }
        l_first = P;
    }
    // (layer 1's fragments are fetched behind layer 0's aggregation like every other layer's, not here: they would be live
    // through that aggregation, whose chains run in double, and push the row-block state into scratch memory.  bfrag is
    // first read in iteration l = 1, after the fetch at the end of iteration 0.)
    // Layer 0 of a deep stack runs apart from the loop: its aggregation carries the chains in double, and inside the loop
    // the loop-invariant parts of that code (and of the first transform) would be hoisted and kept in registers through all
    // the other layers - the kernel has none to spare (every spill also makes the launch set up scratch memory).
    if (!scores_given && P == 1 && a.num_layers > 1 && a.layers[0].cout == kHid) {
        const FusedLayer& L = a.layers[0];
        const int prio_base = (second && a.prio_second) ? 1 : 0;
        if (a.prio_second || a.prio_gather) set_prio(prio_base);
        first_layer_transform<BLOCK>(a, L, n0 + voff, ng, bufA, bufB, xfill);  // (cluster variant: every row, in every workgroup)
        STAMP(a, g, 3, tclk);
        __syncthreads();
        if (a.prio_gather) set_prio(prio_base + a.prio_gather);
        if constexpr (CLUSTER) {
            cluster_aggregate<BLOCK>(L, bufA, cr, rec, zrow, true);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the "unwritten" marks are in L2 before the next layer starts
        } else {
            hidden_aggregate<BLOCK, kNB>(L, bufA, rb, rec, rinfo, true, nullptr, !a.X && !(MASKED && a.feature_mode == 1));
        }
        STAMP(a, g, 4, tclk);
        // (a wave without a tile - residual graphs late in a search: a handful of tiles on sixteen waves - fetches nothing: every
        // wave pulls the same 8 KB through the L1, 128 KB per layer for a lone 1 024-thread workgroup = ~0.85 us of a layer
        // that computes for ~0.9 us; `tools/stamp_residual.py`)
        if (a.layers[1].cout == kHid && (CLUSTER || wave_has_tile)) load_bfrag(a.layers[1].W, bfrag, DIAG_ON(a, 3), true);  // layer index 1: the f64 MFMA's lane map
        __syncthreads();
        l_first = 1;
    }
    for (int l = l_first; l < (scores_given ? 0 : a.num_layers); ++l) {
        const FusedLayer& L = a.layers[l];
        // fp32 MFMAs and VALU work exclude each other on a SIMD and the older wave wins (tools/micro/mfma_valu.hip):
        // while one workgroup transforms, the other one's gather waves cannot even form their next addresses and the
        // LDS runs dry.  Waves therefore raise their priority for the aggregation phase (a handful of short VALU
        // instructions per LDS round trip) and drop it for the MFMA-paced transform.
        const int prio_base = (second && a.prio_second && (l & 7) < a.prio_second) ? 1 : 0;
        if (a.prio_second || a.prio_gather) set_prio(prio_base);
        if constexpr (CLUSTER) {
            if (l == 1) {  // before the first pull: every workgroup of the graph has marked its exchange rows
                cluster_wait<BLOCK>(xfl, K, a.nonce, a.status);
                if (a.cluster_inject && threadIdx.x == 0 && a.status) atomicOr(a.status, DGCN_FAULT_CLUSTER);
                if ((int)threadIdx.x < K) {  // all on one XCD?  (the cheap visibility rule above depends on it)
                    const int32_t* xcc = xcc_slots;
                    if (__hip_atomic_load(&xcc[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) !=
                        __hip_atomic_load(&xcc[cw], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        if (a.status) atomicOr(a.status, DGCN_FAULT_CLUSTER);
                }
            }
        }
        if (L.cout == kHid) {
#ifdef DGCN_DIAG
            if (a.stamps && stamp_wg && threadIdx.x == 0 && l >= 1 && l <= 8) a.stamps[(size_t)g * 64 + 16 + 4 * (l - 1) + 0] = __builtin_amdgcn_s_memrealtime();
#endif
            if constexpr (CLUSTER) {  // (l >= 1 here: layer 0 ran above)
                float* xz0 = a.xz + (size_t)g * 3 * a.max_nodes * kHid;
                float* slice = xz0 + (size_t)(l % 3) * a.max_nodes * kHid;
                if (l == 1) hidden_transform_owned_f64<BLOCK>(bfrag, ctile, bufA, bufB, slice);
                else hidden_transform_owned<BLOCK>(bfrag, ctile, bufA, bufB, slice);
                cluster_pull_rows<BLOCK>(slice, bufB, perm, ng, K, cw, a.status);
                cluster_mark_unwritten<BLOCK>(ctile, xz0 + (size_t)((l + 2) % 3) * a.max_nodes * kHid);
            }
            else if (l == 1) hidden_transform_f64<BLOCK>(bfrag, ng, bufA, bufB);
            else if (!DIAG_ON(a, 1)) hidden_transform<BLOCK>(bfrag, ng, bufA, bufB);
#ifdef DGCN_DIAG
            if (a.stamps && stamp_wg && threadIdx.x == 0 && l >= 1 && l <= 8) a.stamps[(size_t)g * 64 + 16 + 4 * (l - 1) + 1] = __builtin_amdgcn_s_memrealtime();
#endif
            STAMP(a, g, 5, tclk);  // transform body (wave 0)
            __syncthreads();
            STAMP(a, g, 6, tclk);  // wait at the barrier after transforms
#ifdef DGCN_DIAG
            if (a.stamps && stamp_wg && threadIdx.x == 0 && l >= 1 && l <= 8) a.stamps[(size_t)g * 64 + 16 + 4 * (l - 1) + 2] = __builtin_amdgcn_s_memrealtime();
#endif
            if (a.prio_gather) set_prio(prio_base + a.prio_gather);
            if constexpr (CLUSTER) {
                // with 256 VGPRs the next layer's weight fragments can be requested BEFORE the gather phase: their L2 round
                // trip (~0.8 us, in front of every transform otherwise) hides under it
                if (P == 1 && l + 1 < a.num_layers && a.layers[l + 1].cout == kHid) load_bfrag(a.layers[l + 1].W, bfrag, DIAG_ON(a, 3));
            }
            if constexpr (CLUSTER) {
                cluster_aggregate<BLOCK>(L, bufA, cr, rec, zrow, false);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the "unwritten" marks are in L2 before the next layer starts
            } else {
#ifdef DGCN_DIAG
                if (!DIAG_ON(a, 0)) hidden_aggregate<BLOCK, kNB>(L, bufA, rb, rec, rinfo, false, a.stamps ? a.stamps + (size_t)g * 64 + 48 : nullptr);
#else
                hidden_aggregate<BLOCK, kNB>(L, bufA, rb, rec, rinfo, false);
#endif
            }
#ifdef DGCN_DIAG
            if (a.stamps && stamp_wg && threadIdx.x == 0 && l >= 1 && l <= 8) a.stamps[(size_t)g * 64 + 16 + 4 * (l - 1) + 3] = __builtin_amdgcn_s_memrealtime();
#endif
            STAMP(a, g, 7, tclk);  // gather body (wave 0)
            // fetch the next hidden layer's weights now: they land while this wave waits at the barrier, and
            // their 32 registers are not live during the gather phase
            // (both branches define bfrag: otherwise its 32 registers count as live through the gather of every layer)
            if constexpr (!CLUSTER) {
                if (P == 1 && l + 1 < a.num_layers && a.layers[l + 1].cout == kHid && wave_has_tile) load_bfrag(a.layers[l + 1].W, bfrag, DIAG_ON(a, 3));
                else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
#pragma unroll
                        for (int c = 0; c < 4; ++c) bfrag[i][c] = 0.f;
                }
            }
            __syncthreads();
            STAMP(a, g, 8, tclk);  // wait at the barrier after gathers
        } else {
            // last layer: width 1.  z0 stays in a register, z1 goes to bufB[v] (bufB is free: the
            // previous aggregation finished at the barrier above).
            const int v = threadIdx.x;
            float z0 = 0.f, z1 = 0.f;
            if constexpr (GW) {
                // the words come back - into bufB's tail, clear of the z1 scalars, the priorities and the reduction slots the rest
                // of the kernel keeps at its head (the barrier below, in front of the width-1 aggregation, covers the copy)
                const unsigned rl = rinfo[ng - 1];
                const int used = min(((int)(rl & 0xffff) + (int)(rl >> 16) + 1) >> 1, a.meta_cap >> 1);
                const unsigned* gw = reinterpret_cast<const unsigned*>(a.gwords + (size_t)g * a.meta_cap);
                unsigned* lw = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(bufB) + kGwWordsTail);
                for (int j = threadIdx.x; j < used; j += BLOCK) lw[j] = gw[j];
                words = reinterpret_cast<unsigned short*>(lw);
            }
            // (cluster variant: the last activations exist only for the rows this workgroup owns; the z1 scalars and
            // then the scores go round through the graph's scalar exchange slots)
            bool owned = v < ng;
            if constexpr (CLUSTER) owned = v < ng && ((ipos[v] >> 4) % K) == cw;
            float* xs0 = CLUSTER ? a.xs + (size_t)g * 2 * a.max_nodes : nullptr;
            if (owned) {
                if (P > 1) {  // chains already run block by block above
                    z0 = bufA[v];
                    z1 = bufB[v];
                } else if (l == 1) {  // layer index 1 (a two-layer stack): both chains in double, same k order
                    double d0 = 0.0, d1 = 0.0;
                    for (int k = 0; k < L.cin; ++k) {
                        const double h = (double)bufA[swz(v, k)];
                        d0 = fma(h, (double)L.W[k * 2 + 0], d0);
                        d1 = fma(h, (double)L.W[k * 2 + 1], d1);
                    }
                    z0 = (float)d0;
                    z1 = (float)d1;
                } else if (l > 0 && L.cin == kHid) {  // the row as 8 swizzled 16-byte chunks, same k order
#pragma unroll
                    for (int c = 0; c < kHid / 4; ++c) {
                        const float4 h = *reinterpret_cast<const float4*>(bufA + v * kHid + ((c ^ (v & 7)) << 2));
                        const float hk[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            z0 = fmaf(hk[i], L.W[(4 * c + i) * 2 + 0], z0);
                            z1 = fmaf(hk[i], L.W[(4 * c + i) * 2 + 1], z1);
                        }
                    }
                } else {
                    for (int k = 0; k < L.cin; ++k) {
                        float h;
                        if (l == 0) h = a.X ? a.X[(size_t)(n0 + voff + v) * L.cin + k] : xfill;
                        else h = bufA[swz(v, k)];
                        z0 = fmaf(h, L.W[k * 2 + 0], z0);
                        z1 = fmaf(h, L.W[k * 2 + 1], z1);
                    }
                }
                bufB[v] = z1;
                if constexpr (CLUSTER) xs0[v] = z1;
            }
            if constexpr (CLUSTER) {
                if (v < ng && !owned) bufB[v] = poll_l2_scalar(xs0 + v, a.status);
            }
            __syncthreads();
            if (owned) {
                const unsigned ri = rinfo[v];
                const int rs = ri & 0xffff, re = rs + (ri >> 16);
                float o;
                int j = rs;
                if (l == 0) {  // layer index 0 (a one-layer model): the chain in double, rounded once after + z0 (+ bias)
                    double acc = 0.0;
                    for (; j + 4 <= re; j += 4) {
                        const float a0 = vals[j], a1 = vals[j + 1], a2 = vals[j + 2], a3 = vals[j + 3];
                        const float y0 = bufB[words[j] >> 7], y1 = bufB[words[j + 1] >> 7];
                        const float y2 = bufB[words[j + 2] >> 7], y3 = bufB[words[j + 3] >> 7];
                        acc = fma((double)a0, (double)y0, acc);
                        acc = fma((double)a1, (double)y1, acc);
                        acc = fma((double)a2, (double)y2, acc);
                        acc = fma((double)a3, (double)y3, acc);
                    }
                    for (; j < re; ++j) acc = fma((double)vals[j], (double)bufB[words[j] >> 7], acc);
                    acc = (double)z0 + acc;
                    if (L.bias) acc += (double)L.bias[0];
                    o = (float)acc;
                } else {
                    float acc = 0.f;
                    for (; j + 4 <= re; j += 4) {  // loads of four entries in flight, chain order unchanged
                        const float a0 = vals[j], a1 = vals[j + 1], a2 = vals[j + 2], a3 = vals[j + 3];
                        const float y0 = bufB[words[j] >> 7], y1 = bufB[words[j + 1] >> 7];
                        const float y2 = bufB[words[j + 2] >> 7], y3 = bufB[words[j + 3] >> 7];
                        acc = fmaf(a0, y0, acc);
                        acc = fmaf(a1, y1, acc);
                        acc = fmaf(a2, y2, acc);
                        acc = fmaf(a3, y3, acc);
                    }
                    for (; j < re; ++j) acc = fmaf(vals[j], bufB[words[j] >> 7], acc);
                    o = z0 + acc;
                    if (L.bias) o += L.bias[0];
                }
                score = apply_act(o, L.act);
                if (a.scores && !(CLUSTER && a.do_lgs)) a.scores[n0 + voff + v] = (MASKED && !was_alive) ? 0.f : score;  // (cluster + search: below)
                if constexpr (CLUSTER) xs0[a.max_nodes + v] = score;
            }
            if constexpr (CLUSTER) {
                // the greedy search is the first workgroup's alone: the others have handed in their scores and are done
                if (cw != 0 || !a.do_lgs) {
                    if (fault && a.status) atomicOr(a.status, fault);
                    return;
                }
                if (v < ng && !owned) score = poll_l2_scalar(xs0 + a.max_nodes + v, a.status);
                // every output of the graph leaves from this workgroup: one place to wait for before telling the host
                if (a.scores && v < ng) a.scores[n0 + voff + v] = (MASKED && !was_alive) ? 0.f : score;
            }
            __syncthreads();
            STAMP(a, g, 9, tclk);  // last layer
        }
    }
    if (!a.do_lgs) {
        if (fault && a.status) atomicOr(a.status, fault);
        return;
    }

    // ------------------------------------------------------------ priority + local greedy search
    // heuristics.py:77-116.  Priorities are turned into unique integer ranks under the order
    // (priority desc, index asc) once; a removed vertex gets rank 0xFFFF.  A round is then: every
    // live vertex takes the minimum rank over its adjacency (one LDS read per neighbour), wins iff
    // its own rank is smaller; winners join and kill their neighbours.  lpv lanes share a vertex.
    if constexpr (MASKED) {
        // Residual-graph variant: ranks are taken among the remaining vertices only, removed ones keep
        // their state byte, and one launch is one step of an iterative solver:
        //   greedy_mode 0  `max_rounds` local-greedy rounds   (solve_mwis_dit, mwis_gdpg_call.py:278-318)
        //   greedy_mode 1  the global best joins              (solve_mwis_cit, :343-384)
        //   greedy_mode 2  top-`beam` candidates, each completed greedily by weight; the best total joins
        //                  (solve_mwis_rollout, :596-659)
        double* pr = reinterpret_cast<double*>(bufB);
        double* red = pr + a.max_nodes;          // [BLOCK]
        double* wl = red + BLOCK;          // [max_nodes] vertex weights (rollout totals)
        unsigned short* key = reinterpret_cast<unsigned short*>(bufA);
        unsigned short* gkey = key + a.max_nodes;   // GCN-priority ranks (rollout candidates)
        unsigned short* wkey = gkey + a.max_nodes;  // weight ranks (rollout completions)
        uint8_t* st = reinterpret_cast<uint8_t*>(wkey + a.max_nodes);
        uint8_t* jn = st + a.max_nodes;             // joined flags of one rollout completion
        int* pick = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(bufA) + ((8 * a.max_nodes + 15) & ~15));  // [4]
        double* cand = reinterpret_cast<double*>(pick + 4);                               // [64] candidate totals
        constexpr unsigned kDead = 0xFFFFu;
        const int tv = threadIdx.x;
        int bad = 0;
        double wmine = 0.0;
        if (tv < ng) {
            double p = (double)score;
            wmine = a.weights ? a.weights[n0 + voff + tv] : 0.0;
            if (a.predict_mwis && a.weights) p *= wmine;
            bad = was_alive && (p != p);
            pr[tv] = p;
            wl[tv] = wmine;
            st[tv] = was_alive ? 0 : a.state[n0 + tv];
        }
        // (the rollout: the ranking's counters and the candidate list cleared here, in front of the barrier the NaN vote brings -
        // the step's greedy part is a dozen short phases, and every barrier between two of them costs as much as a phase)
        unsigned* rcnt = reinterpret_cast<unsigned*>(reinterpret_cast<int*>(cand + 64) + 64);  // behind pick / cand / cid
        if (a.greedy_mode == 2) {
            for (int i = threadIdx.x; i < 2 * a.max_nodes + 2; i += BLOCK) rcnt[i] = 0u;
            if (threadIdx.x < 64) reinterpret_cast<int*>(cand + 64)[threadIdx.x] = -1;
        }
        const bool any_bad = block_or<BLOCK>(bad != 0, wflags);
        if (a.greedy_mode != 2) __syncthreads();  // (the vote's flag words are written again by the rounds' votes; the rollout has none)
        if (any_bad) {
            if (threadIdx.x == 0) {
                atomicOr(a.status, fault | DGCN_FAULT_NAN_PRIORITY);
                if (a.rounds) a.rounds[g] = -1;
                if (a.totals) a.totals[g] = 0.0;
            }
            return;
        }
        int lsh = 0;
        while (lsh < 3 && (ng << (lsh + 1)) <= BLOCK) ++lsh;
        const int lpv = 1 << lsh;
        const int vv = threadIdx.x >> lsh, sub = threadIdx.x & (lpv - 1);
        const bool mine = vv < ng;
        // (every vertex of the image is undecided - the renumbering in P0 - so all ng of them are ranked)
        // the candidates: the first `beam` vertices in priority order = ranks 0 .. nc - 1 (every rank below ng exists)
        const int nc = min(min(a.beam, 64), ng);
        unsigned long long* S = reinterpret_cast<unsigned long long*>(rcnt + 2 * a.max_nodes + 2);  // [ng] the completions' state words (below)
        if (a.greedy_mode == 2) {
            rank_blocked<BLOCK, 2, true>(ng, pr, wl, rcnt, a.max_nodes, gkey, wkey);
            if (tv < ng) {  // (this thread's own ranks: written by it just now)
                const unsigned gk = gkey[tv];
                key[tv] = (unsigned short)gk;
                if ((int)gk < nc) reinterpret_cast<int*>(cand + 64)[gk] = tv;  // candidate i is the vertex of rank i: one scatter
                const unsigned kb = (a.options & DGCN_RESIDUAL_COMPLETE_BY_PRIORITY) ? gk : (unsigned)wkey[tv];
                if (nc <= 16) S[tv] = ((unsigned long long)kb << 32) | ((1u << nc) - 1u);
            }
        } else if (a.greedy_mode == 0) {
            rank_blocked<BLOCK, 1>(ng, pr, nullptr, rcnt, a.max_nodes, key, nullptr);
        } else {
            // solve_mwis_cit needs the best vertex only (np.argmax: the first among equals): a reduction, not a ranking
            double bp = tv < ng ? pr[tv] : -1.0 / 0.0;
            int bv = tv < ng ? tv : 0x7fffffff;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const double op = __shfl_xor(bp, off);
                const int ov = __shfl_xor(bv, off);
                if (op > bp || (op == bp && ov < bv)) { bp = op; bv = ov; }
            }
            int* redv = reinterpret_cast<int*>(red + BLOCK / 64);
            if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = bp; redv[threadIdx.x >> 6] = bv; }
            __syncthreads();
            bp = red[0]; bv = redv[0];
#pragma unroll
            for (int w = 1; w < BLOCK / 64; ++w) {
                const double op = red[w];
                const int ov = redv[w];
                if (op > bp || (op == bp && ov < bv)) { bp = op; bv = ov; }
            }
            if (tv < ng) key[tv] = tv == bv ? (unsigned short)0 : (unsigned short)1;  // (greedy_rounds, central: rank 0 joins)
        }
        __syncthreads();
        const int rs = mine ? (int)(rinfo[vv] & 0xffff) : 0, re = mine ? rs + (int)(rinfo[vv] >> 16) : 0;
        int rounds = 0;
        int picked = -1;  // the rollout's pick
        if (a.greedy_mode != 2) {
            rounds = greedy_rounds<BLOCK>(key, st, 2, words, rs, re, vv, sub, lpv, mine, wflags, a.max_rounds, a.greedy_mode == 1);
        } else {
            // candidates in GCN-priority order (stable argsort of -priority = the rank keys): candidate i is the vertex of
            // rank i, so the list is one scatter
            int* cid = reinterpret_cast<int*>(cand + 64);  // [64], filled with the ranks above (fewer remaining vertices than candidates: nc < beam)
            STAMP(a, g, 10, tclk);  // residual step: priorities, ranks, candidates
            // The completions - for each candidate: the residual graph minus its closed neighbourhood, searched greedily by
            // weight (or by priority), total weight of what joins - run CONCURRENTLY, one wave per candidate, each on its own
            // rank array in LDS (bufA / bufB are free here) and without a single workgroup barrier.  A wave walks its instance's
            // vertices in passes of 64 and lets a vertex join as soon as it beats all its live neighbours; removals by earlier
            // passes are visible to later ones, which changes the number of rounds but not the result: the set the local
            // greedy search returns is the unique independent set in which every excluded vertex has a member neighbour ahead
            // of it in the order, whatever the schedule (heuristics.py:13-35 sweeps sequentially, :77-116 in synchronous
            // rounds - same set).  (One after the other with the whole workgroup per candidate this was ~20 barriers x 16
            // candidates = 120 of a step's 300 us at N = 500.)
            if (nc <= 16) {
                // Up to sixteen candidates (the reference's b = 16): ALL completions at once, an instance per bit (the scheme of
                // rollout_bits.h on this kernel's image).  S[v] = live mask | joined mask << 16 | rank << 32; a vertex looks at
                // its neighbours of lower rank: one that has joined kills it (in the instances where it has), none of them left
                // alive lets it join - sixteen instances with a few bit operations per neighbour, no rounds, no barriers: every
                // wave goes over its vertices until they are decided.  Same sets as instance after instance (the set a greedy
                // search by a total order returns does not depend on the schedule, see below); totals in a fixed order.
                // (A wave per candidate on its own rank array, below: 49 us of a step at 500 vertices, 15 at 180.)
                const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
                constexpr int kWaves = BLOCK / 64;
                const unsigned short* kbase = (a.options & DGCN_RESIDUAL_COMPLETE_BY_PRIORITY) ? gkey : wkey;
                volatile unsigned long long* Sv = S;  // (every vertex's word was written with its ranks, in front of the barrier above)
                volatile unsigned* Slo = reinterpret_cast<volatile unsigned*>(S);  // word 2 v: the masks of vertex v
                const unsigned myrank = tv < ng ? (unsigned)kbase[tv] : 0u;
                for (int i = wave; i < nc; i += kWaves) {  // instance i: candidate i and its neighbours do not take part
                    const int c = cid[i];
                    const int crs = (int)(rinfo[c] & 0xffff), cre = crs + (int)(rinfo[c] >> 16);
                    for (int j = crs + lane; j < cre; j += 64)  // (diagonal entry: c itself)
                        atomicAnd(reinterpret_cast<unsigned*>(S) + 2 * (words[j] >> 7), ~(1u << i));
                }
                __syncthreads();
                {
                    const int vrs = tv < ng ? (int)(rinfo[tv] & 0xffff) : 0, vre = tv < ng ? vrs + (int)(rinfo[tv] >> 16) : 0;
                    bool more;
                    do {
                        more = false;
                        unsigned mine = tv < ng ? Slo[2 * tv] : 0u;
                        const unsigned live = mine & 0xffffu;
                        if (live) {
                            unsigned seen = 0u;
                            int j = vrs;
                            for (; j + 3 < vre; j += 4) {  // four word -> state chains in flight
                                const int u0 = words[j] >> 7, u1 = words[j + 1] >> 7, u2 = words[j + 2] >> 7, u3 = words[j + 3] >> 7;
                                const unsigned long long s0 = Sv[u0], s1 = Sv[u1], s2 = Sv[u2], s3 = Sv[u3];
                                seen |= ((unsigned)(s0 >> 32) < myrank ? (unsigned)s0 : 0u) | ((unsigned)(s1 >> 32) < myrank ? (unsigned)s1 : 0u) |
                                        ((unsigned)(s2 >> 32) < myrank ? (unsigned)s2 : 0u) | ((unsigned)(s3 >> 32) < myrank ? (unsigned)s3 : 0u);
                            }
                            for (; j < vre; ++j) {
                                const unsigned long long s0 = Sv[words[j] >> 7];
                                seen |= (unsigned)(s0 >> 32) < myrank ? (unsigned)s0 : 0u;
                            }
                            const unsigned killed = seen >> 16;
                            const unsigned die = live & killed, win = live & ~killed & ~(seen & 0xffffu);
                            if (die | win) {
                                mine = (live & ~die & ~win) | (((mine >> 16) | win) << 16);
                                Slo[2 * tv] = mine;
                            }
                            more = (mine & 0xffffu) != 0u;
                        }
                    } while (__any(more));
                }
                __syncthreads();
                for (int i = wave; i < nc; i += kWaves) {  // totals: instance i on wave i, lane-strided partials + a shuffle tree
                    double tot = 0.0;
                    for (int v = lane; v < ng; v += 64)
                        if ((Slo[2 * v] >> (16 + i)) & 1u) tot += wl[v];
                    tot = wave_sum_f64(tot);
                    if (lane == 0) cand[i] = wl[cid[i]] + tot;
                }
            } else {
                const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
                constexpr int kWaves = BLOCK / 64;
                unsigned short* kbase = (a.options & DGCN_RESIDUAL_COMPLETE_BY_PRIORITY) ? gkey : wkey;
                // per-wave rank arrays behind everything the phase keeps in bufA (keys, states, pick, cand, cid)
                unsigned short* kw0 = reinterpret_cast<unsigned short*>(cid + 64);
                const int kstride = (a.max_nodes + 7) & ~7;
                for (int i = wave; i < nc; i += kWaves) {
                    unsigned short* kw = kw0 + (size_t)wave * kstride;
                    const int c = cid[i];
                    for (int v = lane; v < ng; v += 64) kw[v] = kbase[v];
                    {
                        const int crs = (int)(rinfo[c] & 0xffff), cre = crs + (int)(rinfo[c] >> 16);
                        for (int j = crs + lane; j < cre; j += 64) kw[words[j] >> 7] = (unsigned short)kDead;  // diagonal entry: c itself
                    }
                    double tot = 0.0;
                    while (true) {
                        bool any_live = false;
                        for (int v0 = 0; v0 < ng; v0 += 64) {
                            const int v = v0 + lane;
                            const unsigned kv = v < ng ? (unsigned)kw[v] : kDead;
                            const bool live = kv != kDead;
                            unsigned m = kDead;
                            const int vrs = live ? (int)(rinfo[v] & 0xffff) : 0, vre = live ? vrs + (int)(rinfo[v] >> 16) : 0;
                            // (four word -> rank chains in flight: one after the other, two LDS round trips per neighbour,
                            // this loop was the completions' time - 60 us of a step at 500 vertices)
                            int j = vrs;
                            for (; j + 3 < vre; j += 4) {
                                const int u0 = words[j] >> 7, u1 = words[j + 1] >> 7, u2 = words[j + 2] >> 7, u3 = words[j + 3] >> 7;
                                const unsigned k0 = kw[u0], k1 = kw[u1], k2 = kw[u2], k3 = kw[u3];
                                m = min(min(m, u0 != v ? k0 : kDead), min(u1 != v ? k1 : kDead, min(u2 != v ? k2 : kDead, u3 != v ? k3 : kDead)));
                            }
                            for (; j < vre; ++j) {
                                const int u = words[j] >> 7;
                                const unsigned ku = kw[u];
                                if (u != v) m = min(m, ku);
                            }
                            const bool won = live && kv < m;
                            if (won) {
                                for (int j = vrs; j < vre; ++j) kw[words[j] >> 7] = (unsigned short)kDead;  // neighbours and itself
                                tot += wl[v];
                            }
                            any_live |= live && !won;
                        }
                        if (!__any(any_live)) break;
                    }
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
                    if (lane == 0) cand[i] = wl[c] + tot;
                }
            }
            __syncthreads();
            {
                // np.isclose(cand, cand.max(), rtol=1e-12, atol=0): the first candidate within tolerance wins (none within it -
                // totals that are not finite -: the first).  A lane per candidate, every wave for itself: no barrier, no serial
                // walk of thread 0 over the list.
                const int lane = threadIdx.x & 63;
                const double cv = lane < nc ? cand[lane] : -1.0 / 0.0;
                const double mx = wave_max_f64(cv);
                // (cv == mx: equal infinities are close for np.isclose, fabs(inf - inf) is NaN; nobody close - NaN totals -: the first)
                const unsigned long long tied = __ballot(lane < nc && (cv == mx || fabs(cv - mx) <= 1e-12 * fabs(mx)));
                picked = cid[tied ? __ffsll((long long)tied) - 1 : 0];
            }
            const int c = picked;
            const int crs = (int)(rinfo[c] & 0xffff), cre = crs + (int)(rinfo[c] >> 16);
            for (int j = crs + tv; j < cre; j += BLOCK) {
                const int u = words[j] >> 7;
                st[u] = (u == c) ? 1 : 2;
            }
            rounds = 1;
            __syncthreads();
        }
        if constexpr (CLUSTER) {
            // A placement fault anywhere in this launch (a workgroup that never met its peers: its rows - and whatever was computed
            // from them - are not valid): the step leaves this graph's search AS IT WAS.  The status word says so, the caller
            // switches the variant off and takes the step again (Engine.solve_residual does): a search survives the fault.
            // (Whoever reported the fault did so before handing anything on, so the word is set before this workgroup got here.)
            __syncthreads();
            const bool lost = threadIdx.x == 0 && (__hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & DGCN_FAULT_CLUSTER) != 0;
            if (block_or<BLOCK>(lost, wflags)) return;
            __syncthreads();
        }
        if (tv < ng && was_alive) a.state[n0 + voff + tv] = st[tv];
        if (threadIdx.x == 0 && a.rounds) a.rounds[g] = rounds;
        if (threadIdx.x == 0 && a.progress) atomicAdd(a.progress, 1);
        if (a.totals && picked >= 0) {  // (the rollout: one vertex joined)
            if (threadIdx.x == 0) a.totals[g] = a.weights ? wl[picked] : pr[picked];
        } else if (a.totals) {
            double part = 0.0;
            if (tv < ng && was_alive && st[tv] == 1) part = a.weights ? wl[tv] : pr[tv];
            const double tot = block_sum<BLOCK>(part, red);
            if (threadIdx.x == 0) a.totals[g] = tot;
        }
        if (fault) atomicOr(a.status, fault);
        STAMP(a, g, 11, tclk);  // residual step: rounds / completions + pick, state, totals
        STAMP_FLUSH(a, g);
    } else {
    double* pr = reinterpret_cast<double*>(bufB);
    double* red = pr + a.max_nodes;  // [BLOCK] slots; bufB has 128 B per row and max_nodes >= 64 rows
    unsigned short* key = reinterpret_cast<unsigned short*>(bufA);
    uint8_t* st = reinterpret_cast<uint8_t*>(key + a.max_nodes);
    constexpr unsigned kDead = 0xFFFFu;
    int bad = 0;
    if ((int)threadIdx.x < ng) {
        double p = (double)score;
        if (a.predict_mwis && a.weights) p *= a.weights[n0 + threadIdx.x];
        bad = (p != p);
        pr[threadIdx.x] = p;
    }
    // (the flag words are next written by the first greedy round, behind the barrier that follows the ranks)
    const bool any_bad = block_or<BLOCK>(bad != 0, wflags);
    if (any_bad) {
        if (threadIdx.x == 0) {
            atomicOr(a.status, fault | DGCN_FAULT_NAN_PRIORITY);
            if (a.rounds) a.rounds[g] = -1;
            if (a.totals) a.totals[g] = 0.0;
        }
        if ((int)threadIdx.x < ng) a.state[n0 + threadIdx.x] = 0;
        signal_done(a);
        return;
    }
    int lsh = 0;  // lanes per vertex = 1 << lsh, as many as the block affords (<= 8)
    while (lsh < 3 && (ng << (lsh + 1)) <= BLOCK) ++lsh;
    const int lpv = 1 << lsh;
    const int vv = threadIdx.x >> lsh, sub = threadIdx.x & (lpv - 1);
    const bool mine = vv < ng;
    {
        int cnt = 0;
        if (mine) {
            const double pvv = pr[vv];
#pragma unroll 4
            for (int w = sub; w < ng; w += lpv) {
                const double pw = pr[w];
                cnt += (pw > pvv) || (pw == pvv && w < vv);
            }
        }
        for (int off = 1; off < lpv; off <<= 1) cnt += __shfl_xor(cnt, off);
        if (mine && sub == 0) { key[vv] = (unsigned short)cnt; st[vv] = 0; }
    }
    // (Tried and dropped: rounds with every lane's neighbours as (rank << 16 | vertex) in registers, double-buffered "gone"
    // bytes and one barrier per round, ranks with eight reads in flight.  14 -> 8.5 us when the phase runs alone, but the
    // C3 launch did not move (the co-resident workgroup's gathers keep the LDS queue full: every dependent access costs
    // ~500 cycles either way) and the one-layer configurations lost 10 % to the set-up.)
    int rounds = 0;
    const int rs = mine ? (int)(rinfo[vv] & 0xffff) : 0, re = mine ? rs + (int)(rinfo[vv] >> 16) : 0;
    if constexpr (CLUSTER) {
        // One graph (or a few) with the CU to itself: nothing hides a dependent LDS access here, so the rounds keep every
        // lane's first kNb neighbours in registers (one round trip per round with all reads in flight instead of two
        // dependent ones per four entries); entries past kNb * lpv go through the table as below.  Same rounds, same sets.
        constexpr int kNb = 20;
        const int dummy = (3 * a.max_nodes + 1) / 2 + 2;  // a rank slot nobody owns: always "removed"
        int nb[kNb];
#pragma unroll
        for (int i = 0; i < kNb; ++i) {
            const int j = rs + sub + i * lpv;
            const int u = j < re ? (int)(words[j] >> 7) : vv;
            nb[i] = (j < re && u != vv) ? u : dummy;
        }
        if (threadIdx.x == 0) key[dummy] = (unsigned short)kDead;
        __syncthreads();
        while (true) {
            const unsigned mykey = mine ? (unsigned)key[vv] : kDead;
            const bool live = mykey != kDead;
            unsigned k[kNb];
#pragma unroll
            for (int i = 0; i < kNb; ++i) k[i] = key[nb[i]];
            unsigned m = kDead;
#pragma unroll
            for (int i = 0; i < kNb; ++i) m = min(m, k[i]);
            if (live)
                for (int j = rs + sub + kNb * lpv; j < re; j += lpv) {
                    const int u = words[j] >> 7;
                    const unsigned kk = key[u];
                    if (u != vv) m = min(m, kk);
                }
            for (int off = 1; off < lpv; off <<= 1) m = min(m, (unsigned)__shfl_xor((int)m, off));
            const bool won = live && mykey < m;
            if (!block_or<BLOCK>(live, wflags)) break;  // its barrier also orders every rank read before the kills below
            ++rounds;
            if (won) {
#pragma unroll
                for (int i = 0; i < kNb; ++i) { key[nb[i]] = (unsigned short)kDead; st[nb[i]] = 2; }  // (the nobody-slot takes its share)
                for (int j = rs + sub + kNb * lpv; j < re; j += lpv) {
                    const int u = words[j] >> 7;
                    if (u != vv) { key[u] = (unsigned short)kDead; st[u] = 2; }
                }
                if (sub == 0) { key[vv] = (unsigned short)kDead; st[vv] = 1; }
            }
            __syncthreads();
        }
    } else {
    __syncthreads();
    while (!DIAG_ON(a, 2)) {
        const unsigned mykey = mine ? (unsigned)key[vv] : kDead;
        const bool live = mykey != kDead;
        unsigned m = kDead;
        if (live) {
            int j = rs + sub;
            for (; j + 3 * lpv < re; j += 4 * lpv) {  // four independent word -> rank chains in flight
                const int u0 = words[j] >> 7, u1 = words[j + lpv] >> 7, u2 = words[j + 2 * lpv] >> 7,
                          u3 = words[j + 3 * lpv] >> 7;
                const unsigned k0 = key[u0], k1 = key[u1], k2 = key[u2], k3 = key[u3];
                m = min(m, u0 != vv ? k0 : kDead);
                m = min(m, u1 != vv ? k1 : kDead);
                m = min(m, u2 != vv ? k2 : kDead);
                m = min(m, u3 != vv ? k3 : kDead);
            }
            for (; j < re; j += lpv) {
                const int u = words[j] >> 7;
                const unsigned k = key[u];
                if (u != vv) m = min(m, k);
            }
        }
        for (int off = 1; off < lpv; off <<= 1) m = min(m, (unsigned)__shfl_xor((int)m, off));
        const bool won = live && mykey < m;
        if (!block_or<BLOCK>(live, wflags)) break;  // its barrier also orders every rank read before the kills below
        ++rounds;
        if (won) {
            for (int j = rs + sub; j < re; j += lpv) {
                const int u = words[j] >> 7;
                if (u != vv && key[u] != kDead) { key[u] = (unsigned short)kDead; st[u] = 2; }
            }
            if (sub == 0) { key[vv] = (unsigned short)kDead; st[vv] = 1; }
        }
        __syncthreads();
    }
    }
    STAMP(a, g, 10, tclk);  // priorities, ranks, greedy rounds
    const int tv = threadIdx.x;
    if (tv < ng) a.state[n0 + tv] = st[tv];
    if (threadIdx.x == 0 && a.rounds) a.rounds[g] = rounds;
    if (a.totals) {
        double part = 0.0;
        if (tv < ng && st[tv] == 1) part = a.weights ? a.weights[n0 + tv] : pr[tv];
        const double tot = block_sum<BLOCK>(part, red);
        if (threadIdx.x == 0) a.totals[g] = tot;
    }
    if (fault) atomicOr(a.status, fault);
    signal_done(a);
    STAMP(a, g, 11, tclk);  // totals, output
    STAMP_FLUSH(a, g);
    }
}

// ---------------------------------------------------------------------------------------------
// entry slots: the entries themselves plus at most one padding slot per row (even row starts)
// Block-major records of a graph (row_blocks_init): 64 ceil(c_b / 4) <= 16 (c_b + 3) for block b, whose first row has c_b
// entries; c_0 <= N, and 16 c_b for b >= 1 is at most what the 16 rows of block b - 1 hold together (descending order),
// so everything stays under 16 N + entries + 3 (N + 15), + 64 for the load one trip ahead (+ 16 per block for the one
// trip a block of empty rows still makes).  meta_cap covers the entries.
static int fused_rec_cap(int meta_cap, int max_nodes) { return meta_cap + ((20 * max_nodes + 192 + 15) & ~15); }
static int fused_meta_cap(int max_graph_nnz, int max_nodes) { return (max_graph_nnz + max_nodes + 2 + 16 + 15) & ~15; }  // 16 records = 128 B: slices never share a cache line

static size_t fused_lds_bytes(int max_nodes, int meta_cap, bool gvals) {
    const size_t bufs = (size_t)max_nodes * kHid * sizeof(float) * 2;
    const size_t rinfo = (size_t)((max_nodes + 3) & ~3) * sizeof(unsigned);
    return ((bufs + rinfo + (size_t)meta_cap * (gvals ? 2 : 6) + (size_t)max_nodes * 4 + 127) & ~(size_t)127) + 256;  // + block-OR flags (128 B) + a zero row (128 B)
}

constexpr size_t kLdsLimit = 160 * 1024;

// k_fused<.., GW>: bufA, bufB, the row tables, the flag words - nothing per entry
static size_t fused_lds_bytes_gw(int max_nodes) {
    const size_t bufs = (size_t)max_nodes * kHid * sizeof(float) * 2;
    const size_t rinfo = (size_t)((max_nodes + 3) & ~3) * sizeof(unsigned);
    return ((bufs + rinfo + (size_t)max_nodes * 4 + 127) & ~(size_t)127) + 256;
}

// 0: everything in LDS; 1: entry values in global scratch; -1: the image does not fit either way
static int fused_variant(int max_nodes, int meta_cap) {
    if (fused_lds_bytes(max_nodes, meta_cap, false) <= kLdsLimit) return 0;
    if (fused_lds_bytes(max_nodes, meta_cap, true) <= kLdsLimit) return 1;
    return -1;
}

static int device_cus();
// Does the words-in-global variant take this (batch, model)?  Plain solves only (the caller knows), a batch of more graphs
// than CUs (otherwise every graph gets a CU to itself anyway, and sixteen waves serve it better), an image that does NOT leave
// two workgroups per CU as it is but does so without its entry arrays, graphs of at most 384 vertices, and room for the words
// in the buffers' space before and behind the layers.
static bool fused_gw_takes(const DgcnBatch* b, const DgcnModel* m, int meta_cap) {
    const int want = opt(OPT_FUSED_GW);  // -1 automatic, 0 never, 1 wherever it fits (tests)
    if (want == 0 || !m->layers_host || m->num_layers < 2) return false;
    const int mn = max(b->max_nodes, 64);
    if (b->max_nodes > kGwMaxNodes) return false;
    if (fused_lds_bytes_gw(mn) > kLdsLimit / 2) return false;
    if ((size_t)kGwWordsP0 + 2 * (size_t)meta_cap > (size_t)mn * kHid * sizeof(float)) return false;
    if ((size_t)kGwWordsTail + 2 * (size_t)meta_cap > (size_t)mn * kHid * sizeof(float)) return false;
    if (want == 1) return true;
    // Measured on the BA test mix (100 .. 300 vertices, 20 layers; tools/ab_fused.py, profiles/r06_fused_gw.txt), 1 024-thread launch ->
    // this variant: 500 graphs 299 -> 341 us (every workgroup resident at once: the launch is as long as its largest graph, and that
    // one is served by eight waves instead of sixteen), 768 graphs 418 -> 360, 1 000: 548 -> 469, 2 000: 1 071 -> 839, 4 000:
    // 2 116 -> 1 671 - it pays once the CUs are refilled as workgroups finish: from three graphs per CU on.
    if ((long)b->num_graphs < 3L * device_cus()) return false;
    return fused_lds_bytes(mn, meta_cap, false) > kLdsLimit / 2;  // (an image that shares a CU as it is stays as it is)
}

static int device_cus() {
    static std::atomic<int> cu_count[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int ncu = cu_count[dev & 63].load(std::memory_order_relaxed);
    if (ncu == 0) {
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) ncu = 1;
        cu_count[dev & 63].store(ncu, std::memory_order_relaxed);
    }
    return ncu;
}

constexpr int kMaxWidePasses = 4;

// F -> c -> 1 with 32 < c <= 128: the first layer runs as ceil(c / 32) column blocks through the 32-wide machinery
static int fused_wide_passes(const DgcnModel* m) {
    if (m->num_layers != 2) return 1;
    const int c = m->layers_host[0].out_dim;
    if (c <= kHid || c > kHid * kMaxWidePasses || m->layers_host[1].out_dim != 1) return 1;
    return (c + kHid - 1) / kHid;
}

static int fused_shape_ok(const DgcnModel* m) {
    if (m->num_layers > kMaxFusedLayers) return 0;
    const int Lc = m->num_layers;
    if (fused_wide_passes(m) > 1) return m->layers_host[0].in_dim <= 64 && m->layers_host[1].in_dim == m->layers_host[0].out_dim;
    for (int l = 0; l < Lc; ++l) {
        const DgcnLayer& L = m->layers_host[l];
        const bool last = l == Lc - 1;
        if (last) { if (L.out_dim != 1) return 0; }
        else if (L.out_dim < 1 || L.out_dim > kHid) return 0;
        if (l > 0 && L.in_dim != m->layers_host[l - 1].out_dim) return 0;
        if (l == 0 && L.in_dim > 64) return 0;
    }
    return 1;
}

// ---- hidden widths other than 32 ----------------------------------------------------------------
// k_fused is written for 32-wide hidden states.  A narrower stack runs as a 32-wide one whose extra weight
// rows / columns (and biases) are zero: the extra features are exactly 0 in every layer and the extra chain
// terms are fmaf(0, 0, acc) = acc, so the real features keep their bits.  A two-layer stack F -> c -> 1 with
// 32 < c <= 128 (the shipped c48 / c64 l=2 checkpoints) runs its first layer as ceil(c / 32) independent
// 32-column blocks - the columns of a GraphConvolution do not interact - while the last layer's two chains
// continue from block to block in the same k order as the one-pass product.  k_pad_model writes the padded /
// re-blocked copies into the caller's workspace (one tiny launch per call; nothing is cached between calls).
constexpr size_t kPadLayerFloats = 64 * 64 + 64;  // [in <= 64][2 * 32] (or [in <= 128][2]) weights + 32 bias (+ slack), per layer

struct PadArgs {
    int32_t num_layers;  // kernel-side ("virtual") layers
    float* out;          // [num_layers][kPadLayerFloats]
    // virtual layer = columns [c0, c0 + out_p) of source layer (W [in][2*out], bias [out]), written as [in_p][2*out_p]
    struct { const float* W; const float* bias; int32_t in, out, c0, in_p, out_p; } src[kMaxFusedLayers];
};

__global__ void k_pad_model(PadArgs a) {
    const int l = blockIdx.x;
    const int in = a.src[l].in, out = a.src[l].out, c0 = a.src[l].c0;
    const int in_p = a.src[l].in_p, out_p = a.src[l].out_p;
    float* W = a.out + (size_t)l * kPadLayerFloats;
    float* bias = W + 64 * 64;
    for (int i = threadIdx.x; i < in_p * 2 * out_p; i += blockDim.x) {
        const int k = i / (2 * out_p), j = i % (2 * out_p);
        const int half = j / out_p, c = c0 + j % out_p;
        W[i] = (k < in && c < out) ? a.src[l].W[k * 2 * out + half * out + c] : 0.f;
    }
    for (int c = threadIdx.x; c < kHid; c += blockDim.x)
        bias[c] = (a.src[l].bias && c < out_p && c0 + c < out) ? a.src[l].bias[c0 + c] : 0.f;
}

static bool fused_needs_padding(const DgcnModel* m) {
    for (int l = 0; l + 1 < m->num_layers; ++l)
        if (m->layers_host[l].out_dim != kHid) return true;
    return false;
}

static int fused_virtual_layers(const DgcnModel* m) { return m->num_layers - 1 + fused_wide_passes(m); }

static size_t fused_pad_bytes(const DgcnModel* m) {
    return fused_needs_padding(m) ? (size_t)fused_virtual_layers(m) * kPadLayerFloats * sizeof(float) : 0;
}

// ---- largest graphs first ---------------------------------------------------------------------
// Workgroups are dispatched in block order and a graph's time grows with its size (transform ~ vertices, aggregation ~
// entries).  A mixed batch that needs more than one round of workgroups (BA test2 mix: 100..300 vertices, 392..11 200
// entries) ends with whatever large graph happened to come late: 418 us per 500-graph launch as the graphs come, 301 us
// largest first, 358 us smallest first (tools/order_probe.py).  One tiny launch turns sizes into a dispatch order:
// key = entries + 16 * vertices (the ratio of the two phases' costs), position = rank under (key desc, index asc).
// key and index in one word, larger = earlier: (entries + 16 * vertices) << gbits | (all ones - index)
__device__ __forceinline__ unsigned graph_key(const int32_t* graph_ptr, const int32_t* row_ptr, const int32_t* edge_ptr, int g, int gbits,
                                              unsigned kmax) {
    const int n0 = graph_ptr[g], n1 = graph_ptr[g + 1];
    const unsigned entries = edge_ptr ? (unsigned)(edge_ptr[g + 1] - edge_ptr[g]) : (unsigned)(row_ptr[n1] - row_ptr[n0]);
    // (clamped to what the batch descriptor promises: whatever the data, the result is a permutation)
    const unsigned key = min(entries + 16u * (unsigned)(n1 - n0), kmax);
    return (key << gbits) | (((1u << gbits) - 1u) - (unsigned)g);
}

// Sixteen lanes per graph: a graph's rank is a count over ALL keys, and a batch of 4 000 graphs dealt a thread per graph kept 16 workgroups
// busy for 47 us (2.7 % of C4's step; 7.5 us for 500 graphs).  A block ranks 16 graphs; every lane scans a sixteenth of each tile.
constexpr int kRankBlock = 256, kRankTile = 1024, kRankLanes = 16, kRankGraphs = kRankBlock / kRankLanes;
__global__ __launch_bounds__(kRankBlock) void k_graph_rank(const int32_t* __restrict__ graph_ptr, const int32_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ edge_ptr, int B, int gbits, unsigned kmax,
                                                            int fold, int32_t* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) unsigned tile[kRankTile];
    const int part = threadIdx.x & (kRankLanes - 1);
    const int g = blockIdx.x * kRankGraphs + (int)(threadIdx.x / kRankLanes);
    const unsigned kg = g < B ? graph_key(graph_ptr, row_ptr, edge_ptr, g, gbits, kmax) : 0u;
    int pos = 0;
    for (int base = 0; base < B; base += kRankTile) {  // (every workgroup works out all keys itself: one launch, not two)
        unsigned ku[kRankTile / kRankBlock];
#pragma unroll
        for (int i = 0; i < kRankTile / kRankBlock; ++i) {  // independent loads: two dependent round trips for the whole tile
            const int u = base + i * kRankBlock + threadIdx.x;
            ku[i] = u < B ? graph_key(graph_ptr, row_ptr, edge_ptr, u, gbits, kmax) : 0u;  // (padding: a key nobody is behind)
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kRankTile / kRankBlock; ++i) tile[i * kRankBlock + threadIdx.x] = ku[i];
        __syncthreads();
        const int m = min(kRankTile, B - base);
        for (int u = 4 * part; u < m; u += 4 * kRankLanes) {
            const uint4 q = *reinterpret_cast<const uint4*>(tile + u);
            pos += (q.x > kg) + (q.y > kg) + (q.z > kg) + (q.w > kg);
        }
    }
#pragma unroll
    for (int d = 1; d < kRankLanes; d <<= 1) pos += __shfl_xor(pos, d);
    // `fold` > 0: the positions from `fold` on run smallest first (fused_fold_at)
    if (g < B && part == 0) order[fold > 0 && pos >= fold ? fold + (B - 1 - pos) : pos] = g;
}

// bits for the graph index in the packed key, or -1 if key and index do not fit one word
static int fused_order_bits(const DgcnBatch* b) {
    int gbits = 1;
    while ((1L << gbits) < b->num_graphs) ++gbits;
    const long key_max = (long)b->max_graph_edges + 16L * b->max_nodes;
    return (key_max << gbits) < (1L << 32) ? gbits : -1;
}

// worth it?  only the host-side shape is known here: more graphs than CUs (several rounds) and a largest graph well above the mean
static bool fused_wants_order(const DgcnBatch* b) {
    if (fused_order_bits(b) < 0) return false;
    if (const int want = opt(OPT_FUSED_ORDER); want >= 0) return want != 0 && b->num_graphs > 1;
    if (b->num_graphs <= device_cus()) return false;
    return (double)b->max_nodes * b->num_graphs > 1.25 * (double)b->num_nodes ||
           (double)b->max_graph_edges * b->num_graphs > 1.25 * (double)b->num_edges;
}

// One-round launches with two workgroups per CU (CUs < graphs <= 2 x CUs): every workgroup is resident from the start, the first
// `CUs` of the dispatch order take a CU each and the rest join them in the same order, so position i shares its CU with position
// CUs + i.  Largest first through BOTH halves pairs the largest graph with the median one; with the second half smallest first
// the largest graph's partner is the smallest, which leaves early and leaves it the CU.  Measured (tools/ab_fused.py, 20 layers,
// profiles/r06_fold.txt): ER(n, 0.1) with n = 80 .. 200 mixed, 500 graphs 173.0 -> 157.4 us, 400: 166.5 -> 156.8, 320: 160.1 -> 157.7;
// the BA mix forced onto k_fused<.., GW>, 500 graphs: 341 -> 310 us (fold points 244 / 250 / 256: the CU count is the best).
// Batches of more than two graphs per CU refill their CUs as workgroups finish: largest first all the way is right there.
// -> position from which the order runs ascending, 0 = plain largest first.  Option "fused_fold": -1 automatic, 0 off, N forced.
static int fused_fold_at(const DgcnBatch* b, bool two_per_cu) {
    const int want = opt(OPT_FUSED_FOLD);
    if (want == 0) return 0;
    if (want > 0) return min(want, b->num_graphs);
    return two_per_cu ? device_cus() : 0;
}

// The variant's switch: option "fused_cluster" (dgcn_set_option; dgcn_set_cluster is the older name of the same word) - set by
// the fault path of host_solver.hip and by the Python engine when a launch reports DGCN_FAULT_CLUSTER.
int cluster_setting() { return max(opt(OPT_FUSED_CLUSTER), -1); }

// How many workgroups per graph (cluster variant of the kernel)?  0 = the ordinary one-workgroup-per-graph launch.
// Only batches so small that CUs would stay idle otherwise: every workgroup of every graph must be resident at once
// (they wait for each other), so graphs (in groups of 8) x K may not exceed the CU count.
// `masked`: the residual-graph launch (a step of an iterative solver).
static int fused_cluster_k(const DgcnBatch* b, const DgcnModel* m, int meta_cap, bool off, bool masked = false) {
    if (off || !m->layers_host || m->num_layers < 2 || fused_wide_passes(m) > 1) return 0;
    if (fused_variant(max(b->max_nodes, 64), meta_cap) < 0 || b->max_nodes > kFusedBlock) return 0;  // (a vertex per thread in the last layer)
    const int blocks = (b->max_nodes + 15) / 16;
    const int gpad = (b->num_graphs + 7) & ~7;
    // every workgroup of every graph must be resident at once; two tiles per workgroup is as fine as it pays
    int K = min(min(8, device_cus() / max(gpad, 1)), (blocks + 1) / 2);
    bool forced = false;
    const int want = cluster_setting();  // -1 automatic, 0 / 1 off, K forced
    if (want >= 0) {
        if (want <= 1) return 0;
        K = min(8, want);
        forced = true;
    }
    if (K < 2 || (long)gpad * K > device_cus() || (blocks + K - 1) / K > kFusedBlock / 64) return 0;  // at most a tile per wave
    // measured (tools/cluster_check.py, 20 layers, one workgroup per graph vs cluster): N = 200: 127 vs 92 us for 1 - 16
    // graphs (K = 6 - 8; 97 with K = 4, 95 with K = 5), 129 vs 94 for 32 (K = 8), 130 vs 100 for 64 (K = 4); N = 300
    // (K = 5+): 126 vs 112; N = 150: 94 vs 82; N = 120: 74 vs 72; N = 77: 61 vs 65.  The fixed cost (every workgroup
    // builds the image, 512 threads for the greedy rounds) is a few us, the gain ~2 us per hidden layer.
    // (tools/cluster_layers.py: N = 200 gains from 4 layers on - 44.6 vs 38.8 us at 5, 59.9 vs 47.5 at 8 -, N = 128 and
    // N = 300 from 7 - 8 on)
    const int min_layers = (blocks >= 10 && blocks <= 16) ? 5 : 8;
    // (five to eight tiles per workgroup run - a tile per wave, two row sets per wave - but do not pay: 64 graphs of N = 500,
    // K = 4: 183.6 against 177.1 us per residual step, 244.6 against 210.4 per rollout step in round 4.  Round 5, with the
    // step's ranking and completions short: complete searches of those 64 graphs 10.7 -> 10.3 ms (rollout), 8.2 -> 7.7 ms (cit)
    // with option "fused_cluster" = 4 - 4 .. 7 %, for a launch that needs EVERY CU of the device free at once (64 x 4 workgroups,
    // one per CU: anything else running makes a workgroup wait for its peers until the spin bound reports a fault).
    // Forced K only: tools/runs/r05_gpu39.sh)
    // (Round 6: the residual launch takes five to eight tiles per workgroup by itself when the batch leaves room for FOUR
    // workgroups per graph - C5's 64 searches on 256 CUs: 10.7 -> 10.3 ms per search in round 5, forced; a search's steps are
    // launched back to back on one stream, nothing else holds its CUs.  A launch whose workgroups do lose each other reports
    // DGCN_FAULT_CLUSTER and the caller's recovery switches the variant off, as for every cluster launch.)
    const int max_tiles = (masked && K >= 4) ? 8 : 4;
    if (!forced && (K < 3 || blocks < 8 || m->num_layers < min_layers || (blocks + K - 1) / K > max_tiles)) return 0;
    return K;
}

static size_t fused_cluster_bytes(const DgcnBatch* b, int K) {
    if (K < 2) return 0;
    const size_t gpad = (size_t)((b->num_graphs + 7) & ~7);
    const size_t mn = (size_t)max(b->max_nodes, 64);
    return 256 + gpad * 8 * (sizeof(unsigned long long) + sizeof(int32_t)) + (size_t)b->num_graphs * mn * (3 * kHid + 2) * sizeof(float) + 256;
}

// Fills the launch arguments shared by both entry points; returns 0 or an error code.
static int fused_prepare(const DgcnBatch* b, const DgcnModel* m, FusedArgs* a, size_t* lds, const char* who,
                         void* workspace, size_t workspace_bytes, bool* gvals, hipStream_t stream, bool no_cluster = false, bool masked = false) {
    if (!fused_shape_ok(m))
        return fail(DGCN_ERR_UNSUPPORTED, "%s: the fused kernel handles F->c->...->c->1 layer stacks with c <= 32 "
                    "(and two-layer stacks F->c->1 with c <= 128) only", who);
    if (b->max_nodes > kFusedMaxNodes)
        return fail(DGCN_ERR_UNSUPPORTED, "%s: graphs of %d vertices exceed the fused kernel's %d", who, b->max_nodes,
                    kFusedMaxNodes);
    a->graph_ptr = b->graph_ptr;
    a->max_nodes = max(b->max_nodes, 64);  // >= 64 rows: the greedy phase re-uses bufB for priorities + reduction
    const int P = fused_wide_passes(m);
    const int VL = fused_virtual_layers(m);
    a->num_layers = VL;
    a->wide_passes = P;
    const size_t pad_bytes = fused_pad_bytes(m);
    if (pad_bytes && (!workspace || workspace_bytes < pad_bytes))
        return fail(DGCN_ERR_ARG, "%s: workspace of %zu bytes needed (zero-padded weights), got %zu", who, pad_bytes,
                    workspace ? workspace_bytes : (size_t)0);
    float* padded = static_cast<float*>(workspace);  // the padded model sits first in the workspace
    PadArgs pa = {};
    pa.num_layers = VL;
    pa.out = padded;
    for (int vl = 0; vl < VL; ++vl) {
        // virtual layer vl <- source layer sl, column block c0
        const int sl = vl < P ? 0 : vl - P + 1;
        const DgcnLayer& L = m->layers_host[sl];
        const bool last = sl == m->num_layers - 1;
        const int c0 = vl < P ? vl * kHid : 0;
        const int in_p = sl == 0 ? L.in_dim : (P > 1 ? kHid * P : kHid), out_p = last ? 1 : kHid;
        a->layers[vl].W = pad_bytes ? padded + (size_t)vl * kPadLayerFloats : L.weights;
        a->layers[vl].bias = pad_bytes ? (L.bias ? padded + (size_t)vl * kPadLayerFloats + 64 * 64 : nullptr) : L.bias;
        a->layers[vl].cin = pad_bytes ? in_p : L.in_dim;
        a->layers[vl].cout = pad_bytes ? out_p : L.out_dim;
        a->layers[vl].act = L.act;
        a->layers[vl].pad = 0;
        pa.src[vl].W = L.weights;
        pa.src[vl].bias = L.bias;
        pa.src[vl].in = L.in_dim;
        pa.src[vl].out = L.out_dim;
        pa.src[vl].c0 = c0;
        pa.src[vl].in_p = in_p;
        pa.src[vl].out_p = out_p;
    }
    if (pad_bytes) {
        TimedLaunch t("fused_pad", stream);
        DGCN_LAUNCH(t, k_pad_model, dim3(VL), dim3(256), 0, stream, pa);
        int rc = check_launch("k_pad_model");
        if (rc) return rc;
        workspace = static_cast<char*>(workspace) + pad_bytes;
        workspace_bytes -= pad_bytes;
    }
    int variant = fused_variant(a->max_nodes, a->meta_cap);
    a->rec_cap = fused_rec_cap(a->meta_cap, b->max_nodes);
    if (variant < 0)
        return fail(DGCN_ERR_UNSUPPORTED, "%s: a graph image of %zu bytes does not fit the 160 KB LDS", who,
                    fused_lds_bytes(a->max_nodes, a->meta_cap, true));
    // plain solves of batches whose images would keep a CU each: values and words out of LDS, two workgroups per CU (k_fused<.., GW>)
    const bool gw = !masked && fused_gw_takes(b, m, a->meta_cap) && fused_cluster_k(b, m, a->meta_cap, no_cluster, masked) < 2;
    if (gw) variant = 2;
    a->gw = gw ? 1 : 0;
    *gvals = variant >= 1;
    if (*gvals) {
        const size_t need = (size_t)b->num_graphs * a->meta_cap * sizeof(float);
        if (!workspace || workspace_bytes < need)
            return fail(DGCN_ERR_ARG, "%s: workspace of %zu bytes needed (entry values of large graphs), got %zu", who,
                        need, workspace ? workspace_bytes : (size_t)0);
        a->gvals = static_cast<float*>(workspace);
        workspace = static_cast<char*>(workspace) + need;
        workspace_bytes -= need;
    }
    if (gw) {
        const size_t need = (((size_t)b->num_graphs * a->meta_cap * sizeof(unsigned short)) + 255) & ~(size_t)255;
        if (!workspace || workspace_bytes < need)
            return fail(DGCN_ERR_ARG, "%s: workspace of %zu bytes needed (gather words of large graphs), got %zu", who,
                        need, workspace ? workspace_bytes : (size_t)0);
        a->gwords = static_cast<unsigned short*>(workspace);
        workspace = static_cast<char*>(workspace) + need;
        workspace_bytes -= need;
    }
    {
        const size_t need = (size_t)b->num_graphs * a->rec_cap * sizeof(uint2) + 256;
        if (!workspace || workspace_bytes < need)
            return fail(DGCN_ERR_ARG, "%s: workspace of %zu bytes needed (entry records), got %zu", who, need,
                        workspace ? workspace_bytes : (size_t)0);
        a->grec = reinterpret_cast<uint2*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
        workspace = static_cast<char*>(workspace) + need;
        workspace_bytes -= need;
    }
    a->num_graphs = b->num_graphs;
    a->order = nullptr;
    // (two 512-thread workgroups per CU, every workgroup of the launch resident at once?  fused_launch_t's choice, seen from here)
    const bool two_per_cu = (gw ? fused_lds_bytes_gw(a->max_nodes) : fused_lds_bytes(a->max_nodes, a->meta_cap, *gvals)) <= kLdsLimit / 2 &&
                            b->num_graphs > device_cus() && b->num_graphs <= 2 * device_cus() && opt(OPT_FUSED_BLOCK) != kFusedBlockBig &&
                            (gw || a->max_nodes <= 16 * kMaxRowBlocks * (kFusedBlock / 64));
    if (fused_wants_order(b)) {
        const size_t need = (size_t)b->num_graphs * sizeof(int32_t) + 256;
        if (!workspace || workspace_bytes < need)
            return fail(DGCN_ERR_ARG, "%s: workspace of %zu bytes needed (dispatch order), got %zu", who, need,
                        workspace ? workspace_bytes : (size_t)0);
        int32_t* order = reinterpret_cast<int32_t*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
        const int blocks = (b->num_graphs + kRankGraphs - 1) / kRankGraphs;
        hipLaunchKernelGGL(k_graph_rank, dim3(blocks), dim3(kRankBlock), 0, stream, b->graph_ptr, a->row_ptr, a->cedge, b->num_graphs,
                           fused_order_bits(b), (unsigned)(b->max_graph_edges + 16 * b->max_nodes), fused_fold_at(b, two_per_cu), order);
        if (int rc = check_launch("k_graph_rank")) return rc;
        a->order = order;
        workspace = static_cast<char*>(workspace) + need;
        workspace_bytes -= need;
    }
    a->cluster = fused_cluster_k(b, m, a->meta_cap, no_cluster, masked);
    if (a->cluster > 1) a->order = nullptr;
    a->cluster_inject = opt(OPT_TEST_CLUSTER_FAULT);
    if (a->cluster > 1) {
        const size_t need = fused_cluster_bytes(b, a->cluster);
        if (!workspace || workspace_bytes < need)
            return fail(DGCN_ERR_ARG, "%s: workspace of %zu bytes needed (exchange buffers), got %zu", who, need,
                        workspace ? workspace_bytes : (size_t)0);
        const size_t gpad = (size_t)((b->num_graphs + 7) & ~7);
        char* p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
        a->xflag = reinterpret_cast<unsigned long long*>(p);
        p += gpad * 8 * (sizeof(unsigned long long) + sizeof(int32_t));
        p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 15) & ~(uintptr_t)15);
        a->xz = reinterpret_cast<float*>(p);
        a->xs = a->xz + (size_t)b->num_graphs * 3 * a->max_nodes * kHid;
        // The workspace is the caller's: between two cluster launches anything may have written these bytes.  Nothing is
        // cleared; a launch's progress words are recognised by a 64-bit nonce no other launch of the process shares
        // (a random start, then a counter): see cluster_wait.
        static std::atomic<unsigned long long> nonce_src{[] {
            std::random_device rd;
            return ((unsigned long long)rd() << 32) ^ (unsigned long long)rd() ^ 0x9E3779B97F4A7C15ull;
        }()};
        a->nonce = nonce_src.fetch_add(0x9E3779B97F4A7C15ull, std::memory_order_relaxed);
    }
    *lds = gw ? fused_lds_bytes_gw(a->max_nodes) : fused_lds_bytes(a->max_nodes, a->meta_cap, *gvals);
    a->flags_off = (int32_t)(*lds - 256);
    return DGCN_OK;
}

template <bool MASKED, bool GVALS, int BLOCK, bool COMPACT = false, bool GW = false>
static int fused_launch_b(FusedArgs& a, int B, size_t lds, const char* family, hipStream_t s) {
    if (lds > 64 * 1024) {
        // raise the kernel's dynamic-LDS limit once per device and size (the attribute call is a driver round trip)
        static std::atomic<size_t> reserved[64];
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::atomic<size_t>& have = reserved[dev & 63];
        if (lds > have.load(std::memory_order_relaxed)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused<MASKED, GVALS, BLOCK, false, COMPACT, GW>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit);
            if (e != hipSuccess) return fail(DGCN_ERR_LAUNCH, "k_fused: cannot reserve %zu bytes of LDS", lds);
            have.store(kLdsLimit, std::memory_order_relaxed);
        }
    }
    // (the compact image build and the residual renumbering keep a vertex per thread - crs / cdg, vid = orig[threadIdx.x]:
    // correct only while a graph has no more vertices than the workgroup has threads; fused_launch_t's block choice
    // guarantees it today, this check keeps a future override from producing wrong rows silently)
    if ((MASKED || COMPACT) && a.max_nodes > BLOCK)
        return fail(DGCN_ERR_LAUNCH, "k_fused: %d vertices per graph on %d threads in a variant that keeps a vertex per thread", a.max_nodes, BLOCK);
    TimedLaunch t(family, s);
    DGCN_LAUNCH(t, (k_fused<MASKED, GVALS, BLOCK, false, COMPACT, GW>), dim3(B), dim3(BLOCK), lds, s, a);
    return check_launch("k_fused");
}

// An image above half the LDS leaves its graph alone on a CU: 16 waves instead of 8 then work on it
// (at 128 VGPRs both fit the register file exactly).  Option "fused_block" = 512 | 1024 overrides (tuning / tests).
template <bool MASKED, bool GVALS, bool COMPACT = false>
static int fused_launch_t(FusedArgs& a, int B, size_t lds, const char* family, hipStream_t s) {
    // ... and so does every graph of a batch that has no more graphs than the device has CUs
    const int ncu = device_cus();
    bool big = ((lds > kLdsLimit / 2 || B <= ncu) && a.max_nodes >= 128) || a.max_nodes > 16 * kMaxRowBlocks * (kFusedBlock / 64);
    if (const int want = opt(OPT_FUSED_BLOCK); want > 0)
        big = (want == kFusedBlockBig && a.max_nodes >= 128) || a.max_nodes > 16 * kMaxRowBlocks * (kFusedBlock / 64);
    return big ? fused_launch_b<MASKED, GVALS, kFusedBlockBig, COMPACT>(a, B, lds, family, s)
               : fused_launch_b<MASKED, GVALS, kFusedBlock, COMPACT>(a, B, lds, family, s);
}

template <bool MASKED, bool GVALS, bool COMPACT = false>
static int fused_launch_cluster(FusedArgs& a, int B, size_t lds, const char* family, hipStream_t s) {
    static std::atomic<size_t> reserved[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > 64 * 1024 && lds > reserved[dev & 63].load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_fused<MASKED, GVALS, kFusedBlock, true, COMPACT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit) != hipSuccess)
            return fail(DGCN_ERR_LAUNCH, "k_fused: cannot reserve %zu bytes of LDS", lds);
        reserved[dev & 63].store(kLdsLimit, std::memory_order_relaxed);
    }
    if (a.max_nodes > kFusedBlock)  // (a vertex per thread in the image build and the last layer: fused_cluster_k checks it too)
        return fail(DGCN_ERR_LAUNCH, "k_fused (cluster): %d vertices per graph on %d threads", a.max_nodes, kFusedBlock);
    TimedLaunch t(family, s);
    DGCN_LAUNCH(t, (k_fused<MASKED, GVALS, kFusedBlock, true, COMPACT>), dim3(((B + 7) & ~7) * a.cluster), dim3(kFusedBlock), lds, s, a);
    return check_launch("k_fused (cluster)");
}

static int fused_launch(FusedArgs& a, int B, size_t lds, const char* family, hipStream_t s, bool masked, bool gvals) {
    a.prio_gather = 1;  // (rounds 2 - 5 swept both: 1 / 5 of every 8 layers; the knobs are gone)
    a.prio_second = 5;
#ifdef DGCN_DIAG
    a.diag = opt(OPT_DIAG_FLAGS);
    a.stamps = reinterpret_cast<unsigned long long*>(static_cast<uintptr_t>(opt64(OPT_DIAG_STAMPS)));
#endif
    const bool compact = a.ccol != nullptr;
    if (compact && masked) return fail(DGCN_ERR_ARG, "k_fused: the residual-graph variant takes expanded batches only");
    if (a.cluster > 1) {
        if (masked) return gvals ? fused_launch_cluster<true, true>(a, B, lds, family, s) : fused_launch_cluster<true, false>(a, B, lds, family, s);
        if (compact) return gvals ? fused_launch_cluster<false, true, true>(a, B, lds, family, s) : fused_launch_cluster<false, false, true>(a, B, lds, family, s);
        return gvals ? fused_launch_cluster<false, true>(a, B, lds, family, s) : fused_launch_cluster<false, false>(a, B, lds, family, s);
    }
    if (a.gw) return compact ? fused_launch_b<false, true, kFusedBlock, true, true>(a, B, lds, family, s)
                             : fused_launch_b<false, true, kFusedBlock, false, true>(a, B, lds, family, s);
    if (masked) return gvals ? fused_launch_t<true, true>(a, B, lds, family, s) : fused_launch_t<true, false>(a, B, lds, family, s);
    if (compact) return gvals ? fused_launch_t<false, true, true>(a, B, lds, family, s) : fused_launch_t<false, false, true>(a, B, lds, family, s);
    return gvals ? fused_launch_t<false, true>(a, B, lds, family, s) : fused_launch_t<false, false>(a, B, lds, family, s);
}

// bytes of global scratch the fused kernel needs for this batch: a token amount when every image fits
// the LDS, otherwise one float per entry slot
static size_t fused_scratch(const DgcnBatch* b, const DgcnModel* m, int meta_cap) {
    size_t need = m->layers_host ? fused_pad_bytes(m) : 0;
    if (fused_variant(max(b->max_nodes, 64), meta_cap) == 1 || (m->layers_host && fused_gw_takes(b, m, meta_cap)))
        need += (size_t)b->num_graphs * meta_cap * (sizeof(float) + sizeof(unsigned short)) + 512;
    need += (size_t)b->num_graphs * fused_rec_cap(meta_cap, b->max_nodes) * sizeof(uint2) + 256;  // entry records of the hidden aggregation
    need += (size_t)b->num_graphs * sizeof(int32_t) + 256;            // dispatch order
    need += fused_cluster_bytes(b, m->layers_host ? max(fused_cluster_k(b, m, meta_cap, false), fused_cluster_k(b, m, meta_cap, false, true)) : 0);
    return need;
}

size_t fused_workspace(const DgcnBatch* b, const DgcnModel* m) {
    return fused_scratch(b, m, fused_meta_cap(b->max_graph_edges + b->max_nodes, b->max_nodes));
}

int fused_forward(const DgcnBatch* b, const DgcnCsr* lap, const DgcnModel* m, const float* X, float x_const,
                  float* scores, void* workspace, size_t workspace_bytes, hipStream_t s) {
    FusedArgs args = {};
    args.row_ptr = lap->row_ptr;
    args.col_idx = lap->col_idx;
    args.vals = lap->values;
    args.from_adj = 0;
    args.meta_cap = fused_meta_cap((lap->max_graph_nnz > 0 ? lap->max_graph_nnz : b->max_graph_edges + b->max_nodes), b->max_nodes);
    args.X = X;
    args.x_const = x_const;
    args.scores = scores;
    args.do_lgs = 0;
    size_t lds = 0;
    bool gvals = false;
    int rc = fused_prepare(b, m, &args, &lds, "dgcn_gcn_forward_batch(mode 1)", workspace, workspace_bytes, &gvals, s,
                           true);  // (no status word on this entry point: no cluster variant, whose placement check reports through it)
    if (rc) return rc;
    return fused_launch(args, b->num_graphs, lds, "fused_forward", s, false, gvals);
}

int shallow_solve(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, const float* X,
                  float x_const, const double* weights, int32_t predict_mwis, float* scores, uint8_t* state, int32_t* rounds,
                  double* totals, int32_t* status, const DoneHook& hook, hipStream_t s);

// general.hip: the same solvers for graphs of any size (compaction + layer-by-layer forward + greedy kernels)
size_t general_workspace(const DgcnBatch* b, const DgcnModel* m);
int general_takes(const DgcnBatch* b, const DgcnModel* m);
int general_setting();
int general_solve(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, const float* X,
                  float x_const, const double* weights, int32_t predict_mwis, float* scores, uint8_t* state, int32_t* rounds,
                  double* totals, int32_t* status, void* workspace, size_t workspace_bytes, hipStream_t s);
int general_residual(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, const float* X,
                     float x_const, int32_t feature_mode, const double* weights, int32_t predict_mwis, int32_t greedy_mode,
                     int32_t max_rounds, int32_t beam, int32_t options, float* scores, uint8_t* state, int32_t* rounds,
                     double* totals, int32_t* progress, int32_t* status, void* workspace, size_t workspace_bytes, hipStream_t s,
                     unsigned long long* tail_word, unsigned long long tail_tag);

// tail.hip: the rest of a graph's search in one launch once at most 64 of its vertices are undecided
int tail_takes(const DgcnModel* m, const float* X, int32_t options);
int tail_finish(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len, float x_const,
                int32_t feature_mode, const double* weights, int32_t predict_mwis, int32_t greedy_mode, int32_t max_rounds,
                int32_t beam, int32_t options, float* scores, uint8_t* state, int32_t* rounds, double* totals, int32_t* progress,
                int32_t* status, hipStream_t s, const unsigned long long* tail_word, unsigned long long tail_tag);
constexpr size_t kTailWordBytes = 512;  // what dgcn_solve_workspace adds behind a path's own scratch: the tail's word, 256-byte aligned

}  // namespace dgcn

using namespace dgcn;

// ---- deep stacks narrower than 32 beyond the fused kernel's sizes ---------------------------------------------------------
// k_big / k_big2 (like k_fused) are written for 32-wide hidden states; the reference also ships c16 / c4 / c3 / c2 deep stacks
// (model/result_IS4SAT_deep_ld1_c16_l20_.., c16_l4, c4_l4, ..: widths from --hidden1, gcn/models.py:550-573) and runs them on
// whatever conflict graph it gets (wireless_dqn_test_mc.py:161: the joint 3 x 300-flow graph has 900 vertices).  Such a model
// runs as its 32-wide zero-padded copy (k_pad_model, above: extra weight rows / columns and biases are zero, the extra
// features are exactly 0 in every layer, the extra chain terms fmaf(0, w, acc) = acc: the real features keep their bits) on
// the one-launch kernels instead of a 20-launch layer-by-layer chain.  The copy sits first in the caller's workspace.
namespace dgcn {
int big_takes(const DgcnBatch* b, const DgcnModel* m);
int big2_takes(const DgcnBatch* b, const DgcnModel* m);
}  // namespace dgcn
struct NarrowModel {
    DgcnModel m;
    DgcnLayer layers[kMaxFusedLayers];
};
static size_t narrow_pad_bytes(const DgcnModel* m) { return ((size_t)m->num_layers * kPadLayerFloats * sizeof(float) + 255) & ~(size_t)255; }
// fills nm with the padded model's descriptors (weights at base + l * kPadLayerFloats); false: not a deep narrow [I, L] stack
static bool narrow_describe(const DgcnModel* m, float* base, NarrowModel& nm) {
    if (!m->layers_host || m->num_supports != 2 || m->num_layers < 3 || m->num_layers > kMaxFusedLayers) return false;
    if (!fused_shape_ok(m) || !fused_needs_padding(m) || fused_wide_passes(m) > 1) return false;
    for (int l = 0; l < m->num_layers; ++l) {
        const DgcnLayer& L = m->layers_host[l];
        if (!L.weights) return false;
        const bool last = l == m->num_layers - 1;
        nm.layers[l].in_dim = l == 0 ? L.in_dim : kHid;
        nm.layers[l].out_dim = last ? 1 : kHid;
        nm.layers[l].weights = base + (size_t)l * kPadLayerFloats;
        nm.layers[l].bias = L.bias ? base + (size_t)l * kPadLayerFloats + 64 * 64 : nullptr;
        nm.layers[l].act = L.act;
    }
    nm.m.num_layers = m->num_layers;
    nm.m.num_supports = 2;
    nm.m.layers_host = nm.layers;
    return true;
}
// ... and only where that is what puts it on a one-launch kernel AND the launch pays (layer by layer the narrow model is the
// cheaper one).  Measured, kernels per call, padded against the chain (profiles/r06_narrow_and_poly.txt: joint 3 x 300 / 3 x 500
// graphs, c16): k_big (up to 976 vertices) is never behind - 32 graphs 331 / 342 us, 128 graphs 343 / 465, 256 graphs 372 / 695 at
// 20 layers, 64 graphs 99 / 99 at four - and every residual step wins (64 graphs: cit 0.185 / 0.290 ms per step, rollout 0.200 /
// 0.363); k_big2's lone 512-thread workgroup per CU needs a batch that fills the device: 64 graphs 974 / 531 us, 128 graphs
// 1 011 / 682, 256 graphs 1 069 / 1 372 - its plain solve is padded from three quarters of the CU count on.
static bool narrow_may_pad(const DgcnBatch* b, const DgcnModel* m, float* base, NarrowModel& nm) {
    return narrow_describe(m, base, nm) && (big_takes(b, &nm.m) || big2_takes(b, &nm.m));
}
static bool narrow_wants_pad(const DgcnBatch* b, const DgcnModel* m, float* base, NarrowModel& nm, bool residual) {
    const int want = opt(OPT_NARROW_PAD);  // (-1 automatic; 0 / 1: the tests' witnesses)
    if (want == 0 || !narrow_may_pad(b, m, base, nm)) return false;
    if (want < 0 && big2_takes(b, &nm.m)) return residual || 4 * (long)b->num_graphs >= 3 * (long)device_cus();
    return true;
}
static int narrow_pad_launch(const DgcnModel* m, const NarrowModel& nm, float* base, hipStream_t stream) {
    PadArgs pa = {};
    pa.num_layers = m->num_layers;
    pa.out = base;
    for (int l = 0; l < m->num_layers; ++l) {
        const DgcnLayer& L = m->layers_host[l];
        pa.src[l].W = L.weights;
        pa.src[l].bias = L.bias;
        pa.src[l].in = L.in_dim;
        pa.src[l].out = L.out_dim;
        pa.src[l].c0 = 0;
        pa.src[l].in_p = nm.layers[l].in_dim;
        pa.src[l].out_p = nm.layers[l].out_dim;
    }
    TimedLaunch t("fused_pad", stream);
    DGCN_LAUNCH(t, k_pad_model, dim3(m->num_layers), dim3(256), 0, stream, pa);
    return check_launch("k_pad_model");
}
// the any-size path's model for this call: the caller's, or its padded copy (written into the head of the workspace, which
// shrinks by that much).  0 or an error code.
static int narrow_swap(const DgcnBatch* b, const DgcnModel*& m, NarrowModel& nm, void*& workspace, size_t& workspace_bytes,
                       const char* who, hipStream_t stream, bool residual) {
    float* base = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    if (!narrow_wants_pad(b, m, base, nm, residual)) return DGCN_OK;
    const size_t need = narrow_pad_bytes(m) + 256;
    if (!workspace || workspace_bytes < need)
        return fail(DGCN_ERR_WORKSPACE, "%s: workspace of %zu bytes needed (dgcn_solve_workspace), got %zu", who, need, workspace ? workspace_bytes : (size_t)0);
    if (int rc = narrow_pad_launch(m, nm, base, stream)) return rc;
    m = &nm.m;
    workspace = reinterpret_cast<char*>(base) + narrow_pad_bytes(m);
    workspace_bytes -= need;
    return DGCN_OK;
}

extern "C" void dgcn_set_cluster(int32_t setting) { opt_store(OPT_FUSED_CLUSTER, setting < -1 ? -1 : setting); }
extern "C" int32_t dgcn_get_cluster(void) { return cluster_setting(); }

extern "C" int dgcn_solve_supported(const DgcnBatch* b, const DgcnModel* m) {
    if (!b || !m || !m->layers_host || m->num_supports != 2) return 0;
    if (!fused_shape_ok(m) || b->max_nodes > kFusedMaxNodes) return 0;
    const int cap = fused_meta_cap(b->max_graph_edges + b->max_nodes, b->max_nodes);
    return fused_variant(max(b->max_nodes, 64), cap) >= 0;
}

// 1 = the per-graph fused kernels (k_fused / k_shallow), 2 = the any-size path of general.hip, 0 = neither
extern "C" int dgcn_solve_path(const DgcnBatch* b, const DgcnModel* m) {
    if (!b || !m || !m->layers_host) return 0;
    if (general_setting() != 1 && dgcn_solve_supported(b, m)) return 1;
    return general_takes(b, m) ? 2 : 0;
}

// a path's own scratch; the residual entry point keeps one more word behind it (DGCN_RESIDUAL_FINISH_SMALL)
static size_t solve_scratch(const DgcnBatch* b, const DgcnModel* m) {
    if (m->layers_host && dgcn_solve_path(b, m) == 2) {
        NarrowModel nm;
        // (one size for both entry points: the padded copy's where either of them would take it)
        if (narrow_may_pad(b, m, reinterpret_cast<float*>(256), nm)) return narrow_pad_bytes(m) + 256 + max(general_workspace(b, &nm.m), general_workspace(b, m));
        return general_workspace(b, m);
    }
    return fused_scratch(b, m, fused_meta_cap(b->max_graph_edges + b->max_nodes, b->max_nodes));
}

extern "C" size_t dgcn_solve_workspace(const DgcnBatch* b, const DgcnModel* m) {
    if (!b || !m) return 0;
    return solve_scratch(b, m) + kTailWordBytes;
}

extern "C" int dgcn_solve_batch(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table, int32_t table_len,
                                const float* X, float x_const, const double* weights, int32_t predict_mwis,
                                float* scores, uint8_t* state, int32_t* rounds, double* totals, int32_t* status,
                                void* workspace, size_t workspace_bytes, void* stream) {
    const DoneHook hook = g_done_hook;  // (host_solver.hip's completion word: this call's, whatever becomes of it)
    g_done_hook = DoneHook{};
    const CompactHook compact = g_compact_hook;  // (... and its compact batch)
    g_compact_hook = CompactHook{};
    if (!b || !m || !m->layers_host || !dinv_table || !state || !status)
        return fail(DGCN_ERR_ARG, "dgcn_solve_batch: null argument");
    if (m->num_supports != 2 && m->num_supports != 3)
        return fail(DGCN_ERR_UNSUPPORTED, "dgcn_solve_batch: [I, L] and [I, L, L.L] models only (num_supports = %d)", m->num_supports);
    if (b->num_graphs <= 0) return DGCN_OK;
    // graphs whose image does not fit a CU's LDS (or models wider than the fused kernel's 32): the any-size path, same results
    if (dgcn_solve_path(b, m) != 1) {
        if (compact.col) return fail(DGCN_ERR_ARG, "dgcn_solve_batch: the any-size path takes expanded batches only");
        NarrowModel nm;  // (deep stacks narrower than 32: zero-padded to 32 where that puts them on k_big / k_big2)
        if (int rc = narrow_swap(b, m, nm, workspace, workspace_bytes, "dgcn_solve_batch", (hipStream_t)stream, false)) return rc;
        return general_solve(b, m, dinv_table, table_len, X, x_const, weights, predict_mwis, scores, state, rounds, totals, status,
                             workspace, workspace_bytes, (hipStream_t)stream);
    }
    // one-layer models: the small dedicated kernel (shallow.hip) - same results, a fraction of the dependent chain
    if (shallow_takes(b, m)) {
        if (compact.col) return fail(DGCN_ERR_ARG, "dgcn_solve_batch: the one-layer kernel takes expanded batches only");
        return shallow_solve(b, m, dinv_table, table_len, X, x_const, weights, predict_mwis, scores, state, rounds, totals, status, hook,
                             (hipStream_t)stream);
    }
    FusedArgs args = {};
    args.row_ptr = b->row_ptr;
    args.col_idx = b->col_idx;
    args.cedge = compact.edge_ptr;
    args.cdeg = compact.deg;
    args.ccol = compact.col;
    args.vals = nullptr;
    args.dinv_table = dinv_table;
    args.table_len = table_len;
    args.from_adj = 1;
    args.meta_cap = fused_meta_cap(b->max_graph_edges + b->max_nodes, b->max_nodes);
    args.X = X;
    args.x_const = x_const;
    args.scores = scores;
    args.weights = weights;
    args.predict_mwis = predict_mwis;
    args.do_lgs = 1;
    args.state = state;
    args.rounds = rounds;
    args.totals = totals;
    args.status = status;
    size_t lds = 0;
    bool gvals = false;
    int rc = fused_prepare(b, m, &args, &lds, "dgcn_solve_batch", workspace, workspace_bytes, &gvals, (hipStream_t)stream);
    if (rc) return rc;
    args.done_flag = hook.flag;
    args.done_count = hook.count;
    args.done_target = hook.target;
    return fused_launch(args, b->num_graphs, lds, "fused_solve", (hipStream_t)stream, false, gvals);
}

extern "C" int dgcn_solve_residual_batch(const DgcnBatch* b, const DgcnModel* m, const double* dinv_table,
                                         int32_t table_len, const float* X, float x_const, int32_t feature_mode,
                                         const double* weights, int32_t predict_mwis, int32_t greedy_mode,
                                         int32_t max_rounds, int32_t beam, int32_t options, float* scores, uint8_t* state,
                                         int32_t* rounds, double* totals, int32_t* progress, int32_t* status,
                                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!b || !m || !m->layers_host || !dinv_table || !state || !status)
        return fail(DGCN_ERR_ARG, "dgcn_solve_residual_batch: null argument");
    if (m->num_supports != 2 && m->num_supports != 3)
        return fail(DGCN_ERR_UNSUPPORTED, "dgcn_solve_residual_batch: [I, L] and [I, L, L.L] models only (num_supports = %d)", m->num_supports);
    if (greedy_mode < 0 || greedy_mode > 2) return fail(DGCN_ERR_ARG, "dgcn_solve_residual_batch: greedy_mode %d", greedy_mode);
    if (greedy_mode == 2 && (beam < 1 || beam > 64 || !weights))
        return fail(DGCN_ERR_ARG, "dgcn_solve_residual_batch: rollout needs weights and 1 <= beam <= 64");
    if ((options & DGCN_RESIDUAL_SCORES_GIVEN) && !scores)
        return fail(DGCN_ERR_ARG, "dgcn_solve_residual_batch: DGCN_RESIDUAL_SCORES_GIVEN needs the scores array");
    if (feature_mode == 1 && (!weights || X))
        return fail(DGCN_ERR_ARG, "dgcn_solve_residual_batch: feature_mode 1 derives X from the weights");
    if (b->num_graphs <= 0) return DGCN_OK;
    // DGCN_RESIDUAL_FINISH_SMALL: after this call's step, graphs with at most 64 undecided vertices run the REST of their
    // search inside one more launch (tail.hip); graphs it does not take go on step by step, call by call
    // The tail pays only when it is the LAST thing a search launches (its one launch runs ~40 steps of every graph side by side;
    // started while other graphs still step call by call it would hold the stream up every time a graph joins it): the step
    // kernels report the largest number of undecided vertices among the active graphs - atomicMax into a word behind the
    // path's scratch, tagged with this call's number so that nothing has to be cleared - and k_tail leaves at once while
    // that is above 64.
    const size_t scratch = solve_scratch(b, m);
    void* const ws_all = workspace;
    const size_t ws_all_bytes = workspace_bytes;
    const bool any_size = dgcn_solve_path(b, m) != 1;
    NarrowModel nm;  // (deep stacks narrower than 32 on the any-size path: zero-padded to 32 where that puts them on k_big / k_big2 - and k_tail)
    if (any_size)
        if (int rc = narrow_swap(b, m, nm, workspace, workspace_bytes, "dgcn_solve_residual_batch", (hipStream_t)stream, true)) return rc;
    bool finish = (options & DGCN_RESIDUAL_FINISH_SMALL) && tail_takes(m, X, options);
    unsigned long long* tail_word = nullptr;
    unsigned long long tail_tag = 0;
    if (finish && ws_all && ws_all_bytes >= scratch + kTailWordBytes) {
        static std::atomic<unsigned long long> calls{1};
        tail_word = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(ws_all) + scratch + 255) & ~(uintptr_t)255);
        tail_tag = calls.fetch_add(1, std::memory_order_relaxed) << 32;
    } else {
        finish = false;
    }
    auto tail = [&](int rc) {
        if (rc != DGCN_OK || !finish) return rc;
        return tail_finish(b, m, dinv_table, table_len, x_const, feature_mode, weights, predict_mwis, greedy_mode, max_rounds, beam,
                           options, scores, state, rounds, totals, progress, status, (hipStream_t)stream, tail_word, tail_tag);
    };
    if (any_size)
        return tail(general_residual(b, m, dinv_table, table_len, X, x_const, feature_mode, weights, predict_mwis, greedy_mode, max_rounds,
                                     beam, options, scores, state, rounds, totals, progress, status, workspace, workspace_bytes,
                                     (hipStream_t)stream, tail_word, tail_tag));
    FusedArgs args = {};
    args.row_ptr = b->row_ptr;
    args.col_idx = b->col_idx;
    args.vals = nullptr;
    args.dinv_table = dinv_table;
    args.table_len = table_len;
    args.from_adj = 1;
    args.meta_cap = fused_meta_cap(b->max_graph_edges + b->max_nodes, b->max_nodes);
    args.X = X;
    args.x_const = x_const;
    args.feature_mode = feature_mode;
    args.scores = scores;
    args.weights = weights;
    args.predict_mwis = predict_mwis;
    args.do_lgs = 1;
    args.greedy_mode = greedy_mode;
    args.max_rounds = max_rounds;
    args.beam = beam;
    args.options = options;
    args.state = state;
    args.rounds = rounds;
    args.totals = totals;
    args.progress = progress;
    args.status = status;
    args.tail_word = tail_word;
    args.tail_tag = tail_tag;
    size_t lds = 0;
    bool gvals = false;
    // (with given scores no layer runs: nothing for a second workgroup to do)
    int rc = fused_prepare(b, m, &args, &lds, "dgcn_solve_residual_batch", workspace, workspace_bytes, &gvals, (hipStream_t)stream,
                           (options & DGCN_RESIDUAL_SCORES_GIVEN) != 0, true);
    if (rc) return rc;
    return tail(fused_launch(args, b->num_graphs, lds, "fused_residual", (hipStream_t)stream, true, gvals));
}
