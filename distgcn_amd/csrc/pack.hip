// Host-side batch ingestion: many per-graph CSR adjacencies -> ONE block-diagonal batch (include/dgcn.h
// "A batch of graphs is ONE block-diagonal CSR") written straight into the caller's (pinned) staging buffer,
// ready for a single host-to-device copy.
//
// The reference parses one .mat file and calls the solver per graph (mwis_dqn_test.py:304-321); the batched
// path needs the graphs side by side.  Doing that with NumPy concatenations costs ~5 ms per 500 graphs - 20x the
// fused kernel's launch time - so it is native: one pass over the inputs, a few worker threads, no
// intermediate copies, structural validation on the way (anything that could make a kernel read out of
// bounds; self-loops / NaNs are data faults the kernels report through the status word).
//
// No device code in this file: it is plain host C++ inside libdgcn.so (built by hipcc like the rest).
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

#include "common.h"

namespace dgcn {

// ---- a small persistent worker pool (thread creation costs more than packing a C3 batch)
// A C3 batch packs in ~0.15 ms and the next one follows within a fraction of a millisecond, so a worker that has just
// finished keeps polling for the next region for a while (kSpin polls) before it sleeps on the condition variable, and
// parts are handed out with a compare-and-swap instead of under the mutex (0.156 ms per C3 batch on 16 threads, 0.16 -
// 0.20 ms with sleeping workers).  A worker that arrives late finds no part left (its region number is stale).
class Pool {
public:
    static Pool& get() {
        static Pool* p = new Pool;  // never destroyed: its (detached) workers may outlive static destruction
        return *p;
    }
    // run fn(part) for part in [0, parts) on up to `parts` threads (the caller takes part 0)
    void run(int parts, const std::function<void(int)>& fn) {
        if (parts <= 1) { fn(0); return; }
        std::unique_lock<std::mutex> call_lock(call_mu_);  // one parallel region at a time
        if (getpid() != pid_) {  // forked child: the parent's worker threads do not exist here
            pid_ = getpid();
            nworkers_ = 0;
            sleepers_.store(0);
        }
        ensure(parts - 1);
        fn_ = &fn;
        parts_.store(parts, std::memory_order_relaxed);
        pending_.store(parts - 1);
        const uint64_t region = (ticket_.load() >> 32) + 1;
        ticket_.store((region << 32) | 1u);  // publishes fn_ / parts_; part 0 is the caller's
        if (sleepers_.load() > 0) {
            std::lock_guard<std::mutex> lk(mu_);
            cv_.notify_all();
        }
        fn(0);
        work(region);  // the caller helps with whatever is left, then waits for the stragglers
        for (unsigned spins = 0; pending_.load() != 0; ++spins) {
            if (spins > 4096) std::this_thread::yield();
            else __builtin_ia32_pause();
        }
    }

private:
    static constexpr int kSpin = 1 << 14;  // ~0.1 - 0.3 ms of polling before a worker sleeps
    Pool() : pid_(getpid()) {}
    void ensure(int n) {
        while (nworkers_ < n) {
            std::thread([this] { loop(); }).detach();
            ++nworkers_;
        }
    }
    // Take parts of region `region` until none are left.  The ticket holds (region, next part): a compare-and-swap that
    // succeeds hands out a part of a region that cannot end before that part is done, so fn_ / parts_ are this region's;
    // a late worker still holding an older region number never changes the ticket.
    void work(uint64_t region) {
        for (;;) {
            uint64_t t = ticket_.load();
            if ((t >> 32) != region) return;
            const int part = (int)(t & 0xffffffffu);
            if (part >= parts_.load(std::memory_order_relaxed)) return;
            if (!ticket_.compare_exchange_weak(t, t + 1)) continue;
            (*fn_)(part);
            pending_.fetch_sub(1);
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            int spins = 0;
            while ((ticket_.load() >> 32) == seen) {
                if (++spins < kSpin) { __builtin_ia32_pause(); continue; }
                std::unique_lock<std::mutex> lk(mu_);
                sleepers_.fetch_add(1);
                cv_.wait(lk, [&] { return (ticket_.load() >> 32) != seen; });
                sleepers_.fetch_sub(1);
            }
            seen = ticket_.load() >> 32;
            work(seen);
        }
    }
    std::mutex call_mu_, mu_;
    std::condition_variable cv_;
    int nworkers_ = 0;
    pid_t pid_;
    // written by run() before the ticket of a new region is stored, read by workers after they have seen that ticket
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> parts_{0}, pending_{0}, sleepers_{0};
    std::atomic<uint64_t> ticket_{0};  // region << 32 | next part
};

static inline int64_t align16(int64_t x) { return (x + 15) & ~(int64_t)15; }

template <typename I>
static int measure_t(const void* const* indptr, const int32_t* num_nodes, int B, int with_weights, DgcnPackInfo* info,
                     int64_t* nnz_out) {
    int64_t n = 0, e = 0;
    int max_n = 0;
    int64_t max_e = 0;
    for (int g = 0; g < B; ++g) {
        const int ng = num_nodes[g];
        if (ng < 0 || !indptr[g]) return fail(DGCN_ERR_ARG, "dgcn_pack: graph %d has no indptr / a negative size", g);
        const I* p = static_cast<const I*>(indptr[g]);
        const int64_t eg = (int64_t)p[ng];
        if (p[0] != 0 || eg < 0) return fail(DGCN_ERR_ARG, "dgcn_pack: graph %d: indptr must start at 0 and end >= 0", g);
        if (nnz_out) nnz_out[g] = eg;
        n += ng;
        e += eg;
        if (ng > max_n) max_n = ng;
        if (eg > max_e) max_e = eg;
    }
    if (n >= (int64_t)1 << 31 || e >= ((int64_t)1 << 31) - n)
        return fail(DGCN_ERR_ARG, "dgcn_pack: batch too large for int32 indices (%lld vertices, %lld entries)", (long long)n, (long long)e);
    info->num_graphs = B;
    info->num_nodes = (int32_t)n;
    info->num_edges = (int32_t)e;
    info->max_nodes = max_n;
    info->max_graph_edges = (int32_t)max_e;
    info->max_degree = 0;
    int64_t off = 0;
    info->off_graph_ptr = off; off = align16(off + (int64_t)(B + 1) * 4);
    info->off_row_ptr = off;   off = align16(off + (n + 1) * 4);
    info->off_col_idx = off;   off = align16(off + (e > 0 ? e : 1) * 4);
    info->off_weights = with_weights ? off : -1;
    if (with_weights) off = align16(off + (n > 0 ? n : 1) * 8);
    info->total_bytes = off > 16 ? off : 16;
    return DGCN_OK;
}

template <typename I>
static int pack_t(const void* const* indptr, const void* const* indices, const double* const* weights,
                  const int32_t* num_nodes, int B, char* dst, DgcnPackInfo* info, int threads, bool reject_self_loops) {
    int32_t* graph_ptr = reinterpret_cast<int32_t*>(dst + info->off_graph_ptr);
    int32_t* row_ptr = reinterpret_cast<int32_t*>(dst + info->off_row_ptr);
    int32_t* col_idx = reinterpret_cast<int32_t*>(dst + info->off_col_idx);
    double* wts = info->off_weights >= 0 ? reinterpret_cast<double*>(dst + info->off_weights) : nullptr;
    // offsets of every graph (sequential: B adds), kept in graph_ptr and a scratch edge-offset array
    std::vector<int32_t> edge_off((size_t)B + 1);
    int64_t n = 0, e = 0;
    for (int g = 0; g < B; ++g) {
        graph_ptr[g] = (int32_t)n;
        edge_off[g] = (int32_t)e;
        n += num_nodes[g];
        e += (int64_t) static_cast<const I*>(indptr[g])[num_nodes[g]];
    }
    graph_ptr[B] = (int32_t)n;
    edge_off[B] = (int32_t)e;
    if (n != info->num_nodes || e != info->num_edges) return fail(DGCN_ERR_ARG, "dgcn_pack_batch: inputs changed since dgcn_pack_measure");
    row_ptr[n] = (int32_t)e;
    if (e == 0) col_idx[0] = 0;
    if (wts && n == 0) wts[0] = 0.0;
    // contiguous graph ranges per worker, balanced on entries + vertices
    threads = std::max(1, std::min(threads, std::min(B, 64)));
    if (e + n < 200000) threads = std::min(threads, 2);  // not worth waking more workers
    std::vector<int> cut((size_t)threads + 1, B);
    cut[0] = 0;
    {
        const int64_t total = e + n;
        int g = 0;
        for (int t = 1; t < threads; ++t) {
            const int64_t target = total * t / threads;
            while (g < B && (int64_t)edge_off[g] + graph_ptr[g] < target) ++g;
            cut[t] = g;
        }
    }
    std::atomic<int> bad_graph{-1};
    std::atomic<int> bad_kind{0};
    std::vector<int> maxdeg((size_t)threads, 0);
    auto work = [&](int part) {
        int md = 0;
        for (int g = cut[part]; g < cut[part + 1]; ++g) {
            const int ng = num_nodes[g];
            const I* p = static_cast<const I*>(indptr[g]);
            const I* c = static_cast<const I*>(indices[g]);
            const int32_t n0 = graph_ptr[g], e0 = edge_off[g];
            const int64_t eg = (int64_t)p[ng];
            if (eg > 0 && !c) { bad_graph = g; bad_kind = 3; continue; }
            int32_t* rp = row_ptr + n0;
            int64_t prev = 0;
            bool ok = true;
            for (int v = 0; v < ng; ++v) {
                const int64_t cur = (int64_t)p[v];
                ok &= cur >= prev;
                const int64_t nxt = (int64_t)p[v + 1];
                const int64_t d = nxt - cur;
                if (d > md) md = (int)std::min<int64_t>(d, 0x7fffffff);
                rp[v] = (int32_t)(e0 + cur);
                prev = cur;
            }
            ok &= eg >= prev;
            if (!ok) { bad_graph = g; bad_kind = 1; continue; }
            int32_t* cc = col_idx + e0;
            unsigned range_bad = 0, self_bad = 0;
            if (reject_self_loops) {  // the plain greedy kernel has no status bit for them and would never finish
                for (int v = 0; v < ng; ++v)
                    for (int64_t j = (int64_t)p[v]; j < (int64_t)p[v + 1]; ++j) {
                        const int64_t u = (int64_t)c[j];
                        range_bad |= (unsigned)(u < 0) | (unsigned)(u >= ng);
                        self_bad |= (unsigned)(u == v);
                        cc[j] = (int32_t)(u + n0);
                    }
            } else {
                for (int64_t j = 0; j < eg; ++j) {
                    const int64_t u = (int64_t)c[j];
                    range_bad |= (unsigned)(u < 0) | (unsigned)(u >= ng);
                    cc[j] = (int32_t)(u + n0);
                }
            }
            if (range_bad) { bad_graph = g; bad_kind = 2; continue; }
            if (self_bad) { bad_graph = g; bad_kind = 5; continue; }
            if (wts) {
                if (!weights || !weights[g]) { if (ng) { bad_graph = g; bad_kind = 4; } continue; }
                std::memcpy(wts + n0, weights[g], (size_t)ng * sizeof(double));
            }
        }
        maxdeg[part] = md;
    };
    Pool::get().run(threads, work);
    if (bad_graph.load() >= 0) {
        static const char* what[] = {"", "indptr is not non-decreasing", "a column index is outside [0, n)", "indices missing",
                                     "weights missing", "the adjacency has a self-loop (heuristics.py:94 would never terminate)"};
        return fail(DGCN_ERR_ARG, "dgcn_pack_batch: graph %d: %s", bad_graph.load(), what[bad_kind.load()]);
    }
    int md = 0;
    for (int t = 0; t < threads; ++t) md = std::max(md, maxdeg[t]);
    info->max_degree = md;
    return DGCN_OK;
}

// ---- compact transfer format (host_solver.hip): what has to cross PCIe is 16-bit LOCAL column ids and a 16-bit degree per
// vertex instead of 32-bit global ids and row pointers - 5.0 MB instead of 9.2 MB for a C3 batch - and the packing loop
// writes half the bytes; k_expand_compact (expand.hip) rebuilds the block-diagonal CSR on the device, entry for entry what
// dgcn_pack_batch writes:
//   [graph_ptr int32[B+1] | edge_ptr int32[B+1] (entries before graph g) | deg uint16[N] | col uint16[E] | weights f64[N]]
// (Also tried: the upper triangle only, 3.0 MB.  The packer then has to split every sorted row at its diagonal and verify
// order and symmetry - 0.74 ms per C3 batch on 8 threads against 0.26 ms for the ordinary format: the host, not PCIe, is
// what the pipeline waits for.)
// Returns DGCN_OK, an error, or 1: a graph has more than 65 535 vertices or a vertex more than 65 535 neighbours - the
// caller then packs the ordinary format.
int compact_layout(const DgcnPackInfo* std_info, DgcnCompactInfo* ci) {
    const int64_t B = std_info->num_graphs, n = std_info->num_nodes, e = std_info->num_edges;
    if (std_info->max_nodes > 65535) return 1;
    int64_t off = 0;
    ci->off_graph_ptr = off; off = align16(off + (B + 1) * 4);
    ci->off_edge_ptr = off;  off = align16(off + (B + 1) * 4);
    ci->off_deg = off;       off = align16(off + (n > 0 ? n : 1) * 2);
    ci->off_col = off;       off = align16(off + (e > 0 ? e : 1) * 2);
    ci->off_weights = std_info->off_weights >= 0 ? off : -1;
    if (std_info->off_weights >= 0) off = align16(off + (n > 0 ? n : 1) * 8);
    ci->total_bytes = off > 16 ? off : 16;
    return 0;
}

template <typename I>
static int pack_compact_t(const void* const* indptr, const void* const* indices, const double* const* weights,
                          const int32_t* num_nodes, int B, char* dst, DgcnPackInfo* info, const DgcnCompactInfo* ci, int threads,
                          bool reject_self_loops) {
    int32_t* graph_ptr = reinterpret_cast<int32_t*>(dst + ci->off_graph_ptr);
    int32_t* edge_ptr = reinterpret_cast<int32_t*>(dst + ci->off_edge_ptr);
    uint16_t* deg = reinterpret_cast<uint16_t*>(dst + ci->off_deg);
    uint16_t* col = reinterpret_cast<uint16_t*>(dst + ci->off_col);
    double* wts = ci->off_weights >= 0 ? reinterpret_cast<double*>(dst + ci->off_weights) : nullptr;
    int64_t n = 0, e = 0;
    for (int g = 0; g < B; ++g) {
        graph_ptr[g] = (int32_t)n;
        edge_ptr[g] = (int32_t)e;
        n += num_nodes[g];
        e += (int64_t) static_cast<const I*>(indptr[g])[num_nodes[g]];
    }
    graph_ptr[B] = (int32_t)n;
    edge_ptr[B] = (int32_t)e;
    if (n != info->num_nodes || e != info->num_edges) return fail(DGCN_ERR_ARG, "dgcn_pack_batch: inputs changed since dgcn_pack_measure");
    if (wts && n == 0) wts[0] = 0.0;
    threads = std::max(1, std::min(threads, std::min(B, 64)));
    if (e + n < 200000) threads = std::min(threads, 2);
    std::vector<int> cut((size_t)threads + 1, B);
    cut[0] = 0;
    {
        const int64_t total = e + n;
        int g = 0;
        for (int t = 1; t < threads; ++t) {
            const int64_t target = total * t / threads;
            while (g < B && (int64_t)edge_ptr[g] + graph_ptr[g] < target) ++g;
            cut[t] = g;
        }
    }
    std::atomic<int> not_compact{0}, bad_graph{-1}, bad_kind{0};
    std::vector<int> maxdeg((size_t)threads, 0);
    auto work = [&](int part) {
        int md = 0;
        for (int g = cut[part]; g < cut[part + 1]; ++g) {
            const int ng = num_nodes[g];
            const I* p = static_cast<const I*>(indptr[g]);
            const I* __restrict c = static_cast<const I*>(indices[g]);
            const int32_t n0 = graph_ptr[g];
            const int64_t eg = (int64_t)p[ng];
            if (eg > 0 && !c) { bad_graph = g; bad_kind = 3; continue; }
            uint16_t* __restrict dg = deg + n0;
            int64_t prev = 0;
            bool ok = true;
            int gmd = 0;
            for (int v = 0; v < ng; ++v) {
                const int64_t cur = (int64_t)p[v], d = (int64_t)p[v + 1] - cur;
                ok &= cur >= prev && d >= 0;
                if (d > gmd) gmd = (int)std::min<int64_t>(d, 0x7fffffff);
                dg[v] = (uint16_t)d;
                prev = cur;
            }
            ok &= eg >= prev;
            if (!ok) { bad_graph = g; bad_kind = 1; continue; }
            if (gmd > md) md = gmd;
            if (gmd > 65535) { not_compact.store(1, std::memory_order_relaxed); continue; }
            uint16_t* __restrict cc = col + edge_ptr[g];
            unsigned range_bad = 0, self_bad = 0;
            if (reject_self_loops) {
                for (int v = 0; v < ng; ++v)
                    for (int64_t j = (int64_t)p[v]; j < (int64_t)p[v + 1]; ++j) {
                        const int64_t u = (int64_t)c[j];
                        range_bad |= (unsigned)(u < 0) | (unsigned)(u >= ng);
                        self_bad |= (unsigned)(u == v);
                        cc[j] = (uint16_t)u;
                    }
            } else {
                for (int64_t j = 0; j < eg; ++j) {
                    const I u = c[j];
                    range_bad |= (unsigned)(u < 0) | (unsigned)(u >= (I)ng);
                    cc[j] = (uint16_t)u;
                }
            }
            if (range_bad) { bad_graph = g; bad_kind = 2; continue; }
            if (self_bad) { bad_graph = g; bad_kind = 5; continue; }
            if (wts) {
                if (!weights || !weights[g]) { if (ng) { bad_graph = g; bad_kind = 4; } continue; }
                std::memcpy(wts + n0, weights[g], (size_t)ng * sizeof(double));
            }
        }
        maxdeg[part] = md;
    };
    Pool::get().run(threads, work);
    if (bad_graph.load() >= 0) {
        static const char* what[] = {"", "indptr is not non-decreasing", "a column index is outside [0, n)", "indices missing",
                                     "weights missing", "the adjacency has a self-loop (heuristics.py:94 would never terminate)"};
        return fail(DGCN_ERR_ARG, "dgcn_pack_batch: graph %d: %s", bad_graph.load(), what[bad_kind.load()]);
    }
    if (not_compact.load()) return 1;
    int md = 0;
    for (int t = 0; t < threads; ++t) md = std::max(md, maxdeg[t]);
    info->max_degree = md;
    return DGCN_OK;
}

int pack_compact(const void* const* indptr_host, const void* const* indices_host, const double* const* weights_host,
                 const int32_t* num_nodes_host, int32_t num_graphs, int32_t index_bytes, void* staging_host, size_t staging_bytes,
                 DgcnPackInfo* info, const DgcnCompactInfo* ci, int32_t num_threads, bool reject_self_loops) {
    if ((int64_t)staging_bytes < ci->total_bytes) return 1;
    char* dst = static_cast<char*>(staging_host);
    if (index_bytes == 4) return pack_compact_t<int32_t>(indptr_host, indices_host, weights_host, num_nodes_host, num_graphs, dst, info, ci, num_threads, reject_self_loops);
    if (index_bytes == 8) return pack_compact_t<int64_t>(indptr_host, indices_host, weights_host, num_nodes_host, num_graphs, dst, info, ci, num_threads, reject_self_loops);
    return 1;
}

}  // namespace dgcn

using namespace dgcn;

extern "C" int dgcn_pack_measure(const void* const* indptr_host, const int32_t* num_nodes_host, int32_t num_graphs,
                                 int32_t index_bytes, int32_t with_weights, DgcnPackInfo* info, int64_t* nnz_out_host) {
    if (!info || num_graphs < 0 || (num_graphs > 0 && (!indptr_host || !num_nodes_host)))
        return fail(DGCN_ERR_ARG, "dgcn_pack_measure: null argument");
    if (index_bytes == 4) return measure_t<int32_t>(indptr_host, num_nodes_host, num_graphs, with_weights, info, nnz_out_host);
    if (index_bytes == 8) return measure_t<int64_t>(indptr_host, num_nodes_host, num_graphs, with_weights, info, nnz_out_host);
    return fail(DGCN_ERR_ARG, "dgcn_pack_measure: index_bytes must be 4 or 8");
}

namespace dgcn {
int pack_batch(const void* const* indptr_host, const void* const* indices_host, const double* const* weights_host,
               const int32_t* num_nodes_host, int32_t num_graphs, int32_t index_bytes, void* staging_host,
               size_t staging_bytes, DgcnPackInfo* info, int32_t num_threads, bool reject_self_loops) {
    if (!info || !staging_host || num_graphs < 0 || (num_graphs > 0 && (!indptr_host || !indices_host || !num_nodes_host)))
        return fail(DGCN_ERR_ARG, "dgcn_pack_batch: null argument");
    if ((int64_t)staging_bytes < info->total_bytes)
        return fail(DGCN_ERR_WORKSPACE, "dgcn_pack_batch: staging buffer %zu < %lld bytes", staging_bytes, (long long)info->total_bytes);
    if (info->num_graphs != num_graphs) return fail(DGCN_ERR_ARG, "dgcn_pack_batch: info is for %d graphs", info->num_graphs);
    if (num_threads <= 0) {
        unsigned hc = std::thread::hardware_concurrency();
        num_threads = (int)std::min(8u, hc ? hc : 1u);
    }
    char* dst = static_cast<char*>(staging_host);
    if (index_bytes == 4)
        return pack_t<int32_t>(indptr_host, indices_host, weights_host, num_nodes_host, num_graphs, dst, info, num_threads,
                               reject_self_loops);
    if (index_bytes == 8)
        return pack_t<int64_t>(indptr_host, indices_host, weights_host, num_nodes_host, num_graphs, dst, info, num_threads,
                               reject_self_loops);
    return fail(DGCN_ERR_ARG, "dgcn_pack_batch: index_bytes must be 4 or 8");
}
}  // namespace dgcn

extern "C" int dgcn_pack_batch(const void* const* indptr_host, const void* const* indices_host,
                               const double* const* weights_host, const int32_t* num_nodes_host, int32_t num_graphs,
                               int32_t index_bytes, void* staging_host, size_t staging_bytes, DgcnPackInfo* info,
                               int32_t num_threads) {
    return dgcn::pack_batch(indptr_host, indices_host, weights_host, num_nodes_host, num_graphs, index_bytes, staging_host,
                            staging_bytes, info, num_threads, false);
}

extern "C" int dgcn_pack_compact_layout(const DgcnPackInfo* info, DgcnCompactInfo* compact) {
    if (!info || !compact) return fail(DGCN_ERR_ARG, "dgcn_pack_compact_layout: null argument");
    return dgcn::compact_layout(info, compact);
}

extern "C" int dgcn_pack_compact_batch(const void* const* indptr_host, const void* const* indices_host,
                                       const double* const* weights_host, const int32_t* num_nodes_host, int32_t num_graphs,
                                       int32_t index_bytes, void* staging_host, size_t staging_bytes, DgcnPackInfo* info,
                                       const DgcnCompactInfo* compact, int32_t num_threads) {
    if (!info || !compact || !staging_host || num_graphs < 0 || (num_graphs > 0 && (!indptr_host || !indices_host || !num_nodes_host)))
        return fail(DGCN_ERR_ARG, "dgcn_pack_compact_batch: null argument");
    if (info->num_graphs != num_graphs) return fail(DGCN_ERR_ARG, "dgcn_pack_compact_batch: info is for %d graphs", info->num_graphs);
    if (num_threads <= 0) {
        unsigned hc = std::thread::hardware_concurrency();
        num_threads = (int)std::min(8u, hc ? hc : 1u);
    }
    // caller errors are errors here; 1 stays what the header says it is ("not compactable": a graph of more than 65 535
    // vertices or a vertex of more than 65 535 neighbours).  The layout must be THIS batch's: a stale DgcnCompactInfo with
    // smaller offsets would let the sections overrun one another.
    if (index_bytes != 4 && index_bytes != 8) return fail(DGCN_ERR_ARG, "dgcn_pack_compact_batch: index_bytes must be 4 or 8");
    DgcnCompactInfo want;
    if (compact_layout(info, &want) != 0) return 1;
    if (want.off_graph_ptr != compact->off_graph_ptr || want.off_edge_ptr != compact->off_edge_ptr || want.off_deg != compact->off_deg ||
        want.off_col != compact->off_col || want.off_weights != compact->off_weights || want.total_bytes != compact->total_bytes)
        return fail(DGCN_ERR_ARG, "dgcn_pack_compact_batch: the DgcnCompactInfo does not belong to this DgcnPackInfo (dgcn_pack_compact_layout)");
    if ((int64_t)staging_bytes < compact->total_bytes)
        return fail(DGCN_ERR_ARG, "dgcn_pack_compact_batch: staging buffer of %zu bytes, %lld needed", staging_bytes, (long long)compact->total_bytes);
    return dgcn::pack_compact(indptr_host, indices_host, weights_host, num_nodes_host, num_graphs, index_bytes, staging_host,
                              staging_bytes, info, compact, num_threads, false);
}
