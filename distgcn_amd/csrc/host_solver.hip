// Host-to-host solver object: per-graph CSR arrays in host memory -> selected sets in host memory, behind two C calls.
//
// The reference's agents are called with ONE graph at a time - `sess.run` per graph in solve_mwis
// (mwis_dqn_call.py:140-143, mwis_gdpg_call.py:211-216), once per time slot in the wireless loop - and its test loop
// feeds a directory of graphs one after the other (mwis_dqn_test.py:304-321).  For such callers the kernel is a
// small part of a call: packing, two copies, a launch and a wait have to be cheap too.  This object keeps `depth`
// slots of pinned staging memory, device buffers, a stream and an event each, so that
//     dgcn_host_solver_submit  = dgcn_pack_batch into pinned memory -> 1 hipMemcpyAsync -> dgcn_solve_batch
//                                -> 1 hipMemcpyAsync back -> event           (returns at once)
//     dgcn_host_solver_result  = wait for the slot's event -> pointers into its pinned result
// (a batch of up to 2 MB skips both copies: the kernel works on the pinned buffers directly)
// with no interpreter, allocator or framework call in between; several slots overlap packing, copies and kernels of
// consecutive batches.  Shapes the fused kernel does not take go through dgcn_solve_batch's any-size path (general.hip),
// same calls, same results.  No device code in this file.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.h"

struct DgcnHostSolver {
    struct Slot {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        hipEvent_t copied = nullptr;  // the batch has arrived on the device (recorded on the shared copy stream)
        void* in_host = nullptr;   // pinned: the packed batch
        void* in_host_dev = nullptr;  // the same bytes as the device addresses them (small batches are read in place)
        void* in_dev = nullptr;
        size_t in_cap = 0;
        void* exp_dev = nullptr;  // compact transfers: the expanded row_ptr | col_idx (expand.hip)
        bool compact_direct = false;  // ... or the batch goes to the fused kernel as it is:
        const int32_t* c_edge = nullptr;
        const unsigned short* c_deg = nullptr;
        const unsigned short* c_col = nullptr;
        size_t exp_cap = 0;
        void* out_host = nullptr;  // pinned: totals | rounds | status | state
        void* out_host_dev = nullptr;
        void* out_dev = nullptr;
        size_t out_cap = 0;
        int cap_nodes = 0, cap_graphs = 0;
        void* ws = nullptr;
        size_t ws_cap = 0;
        size_t off_totals = 0, off_scores = 0, off_rounds = 0, off_status = 0, off_state = 0, out_bytes = 0;
        int num_nodes = 0, num_graphs = 0;
        bool busy = false;
        bool direct = false;  // this batch: the kernel reads / writes the pinned host buffers itself
        DgcnBatch batch;      // what was launched (pointers into in_dev or in_host_dev): kept for a relaunch
        long long off_weights = -1;
        uint32_t* done_count = nullptr;  // device: graphs finished, all batches of this slot (DoneHook, common.h)
        uint32_t done_total = 0;         // what it reads once the current batch is through (0 = no completion word in use)
        uint32_t done_base = 0;          // graphs launched with a completion word so far
    };
    bool lgs_only = false;  // no model: priority = weight, the plain local greedy search (heuristics.py:77-116)
    DgcnModel model;
    std::vector<DgcnLayer> layers;
    const double* table = nullptr;
    int table_len = 0;
    int predict_mwis = 1;
    float x_const = 1.0f;
    int pack_threads = 0;
    int want_scores = 0;
    int device = 0;
    int next = 0;
    // every host-to-device copy goes through ONE stream: copies issued round-robin on several streams ran at half the
    // rate on MI355X (tools/micro/overlap.hip); the slot's own stream waits for `copied` and runs kernel + copy back
    hipStream_t copy_stream = nullptr;
    std::vector<Slot> slots;
};

namespace dgcn {

static size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// the object's buffers and streams live on the device that was current at create: make it current for a call
struct DeviceScope {
    explicit DeviceScope(int want) {
        if (hipGetDevice(&prev) != hipSuccess) prev = want;
        if (prev != want) (void)hipSetDevice(want);
        else prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    int prev = -1;
};

static void free_in(DgcnHostSolver::Slot& s) {
    if (s.in_host) (void)hipHostFree(s.in_host);
    if (s.in_dev) (void)hipFree(s.in_dev);
    s.in_host = s.in_dev = s.in_host_dev = nullptr;
    s.in_cap = 0;
}

static void free_out(DgcnHostSolver::Slot& s) {
    if (s.out_host) (void)hipHostFree(s.out_host);
    if (s.out_dev) (void)hipFree(s.out_dev);
    s.out_host = s.out_dev = s.out_host_dev = nullptr;
    s.out_cap = 0;
}

static int ensure_in(DgcnHostSolver::Slot& s, size_t bytes) {
    if (bytes <= s.in_cap) return DGCN_OK;
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    free_in(s);
    const size_t cap = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&s.in_host, cap, hipHostMallocDefault) != hipSuccess || hipMalloc(&s.in_dev, cap) != hipSuccess ||
        hipHostGetDevicePointer(&s.in_host_dev, s.in_host, 0) != hipSuccess) {
        free_in(s);
        return fail(DGCN_ERR_WORKSPACE, "dgcn_host_solver: cannot allocate %zu bytes of staging memory", cap);
    }
    s.in_cap = cap;
    return DGCN_OK;
}

// result layout, widest type first, every array 16-byte aligned:
// totals f64[B] | scores f32[N] (optional) | rounds i32[B] | status i32 | state u8[N]
static int ensure_out(DgcnHostSolver::Slot& s, int nodes, int graphs, bool want_scores) {
    if (nodes <= s.cap_nodes && graphs <= s.cap_graphs && s.out_dev) return DGCN_OK;
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    free_out(s);
    const int cn = nodes + nodes / 4 + 64, cg = graphs + graphs / 4 + 8;
    s.off_totals = 0;
    s.off_scores = align16((size_t)cg * 8);
    s.off_rounds = s.off_scores + (want_scores ? align16((size_t)cn * 4) : 0);
    s.off_status = s.off_rounds + align16((size_t)cg * 4);
    s.off_state = s.off_status + 16;
    s.out_bytes = s.off_state + align16((size_t)cn);
    if (hipHostMalloc(&s.out_host, s.out_bytes, hipHostMallocDefault) != hipSuccess ||
        hipMalloc(&s.out_dev, s.out_bytes) != hipSuccess || hipHostGetDevicePointer(&s.out_host_dev, s.out_host, 0) != hipSuccess) {
        free_out(s);
        return fail(DGCN_ERR_WORKSPACE, "dgcn_host_solver: cannot allocate %zu bytes of result memory", s.out_bytes);
    }
    if (hipMemset(s.out_dev, 0, s.out_bytes) != hipSuccess) return fail(DGCN_ERR_LAUNCH, "dgcn_host_solver: hipMemset failed");
    s.out_cap = s.out_bytes;
    s.cap_nodes = cn;
    s.cap_graphs = cg;
    return DGCN_OK;
}

static int ensure_exp(DgcnHostSolver::Slot& s, size_t bytes) {
    if (bytes <= s.exp_cap) return DGCN_OK;
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.exp_dev) (void)hipFree(s.exp_dev);
    s.exp_dev = nullptr;
    s.exp_cap = 0;
    const size_t cap = bytes + bytes / 4 + 4096;
    if (hipMalloc(&s.exp_dev, cap) != hipSuccess) return fail(DGCN_ERR_WORKSPACE, "dgcn_host_solver: cannot allocate %zu bytes for the expanded batch", cap);
    s.exp_cap = cap;
    return DGCN_OK;
}

static int ensure_ws(DgcnHostSolver::Slot& s, size_t bytes) {
    if (bytes <= s.ws_cap) return DGCN_OK;
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.ws) (void)hipFree(s.ws);
    s.ws = nullptr;
    s.ws_cap = 0;
    const size_t cap = bytes + bytes / 4 + 256;
    if (hipMalloc(&s.ws, cap) != hipSuccess) return fail(DGCN_ERR_WORKSPACE, "dgcn_host_solver: cannot allocate %zu bytes of scratch", cap);
    s.ws_cap = cap;
    return DGCN_OK;
}

// kernel(s) + copy back of the batch a slot holds, on the slot's stream
static int launch_slot(DgcnHostSolver* h, DgcnHostSolver::Slot& s) {
    const DgcnBatch& b = s.batch;
    char* base = static_cast<char*>(s.direct ? s.in_host_dev : s.in_dev);
    char* ob = static_cast<char*>(s.direct ? s.out_host_dev : s.out_dev);
    const double* wdev = s.off_weights >= 0 ? reinterpret_cast<const double*>(base + s.off_weights) : nullptr;
    int rc;
    if (h->lgs_only)
        rc = dgcn_lgs_batch(&b, wdev, nullptr, nullptr, 0, reinterpret_cast<uint8_t*>(ob + s.off_state),
                            reinterpret_cast<int32_t*>(ob + s.off_rounds), nullptr, nullptr, wdev,
                            reinterpret_cast<double*>(ob + s.off_totals), reinterpret_cast<int32_t*>(ob + s.off_status), s.stream);
    else {
        if (s.compact_direct) {
            g_compact_hook.edge_ptr = s.c_edge;
            g_compact_hook.deg = s.c_deg;
            g_compact_hook.col = s.c_col;
        }
        if (s.done_total) {  // the finishing workgroups count themselves; the last one writes the word beside the status
            g_done_hook.flag = reinterpret_cast<int32_t*>(ob + s.off_status + 8);
            g_done_hook.count = s.done_count;
            g_done_hook.target = s.done_total;
        }
        rc = dgcn_solve_batch(&b, &h->model, h->table, h->table_len, nullptr, h->x_const, wdev, h->predict_mwis,
                              h->want_scores ? reinterpret_cast<float*>(ob + s.off_scores) : nullptr,
                              reinterpret_cast<uint8_t*>(ob + s.off_state), reinterpret_cast<int32_t*>(ob + s.off_rounds),
                              reinterpret_cast<double*>(ob + s.off_totals), reinterpret_cast<int32_t*>(ob + s.off_status), s.ws,
                              s.ws_cap, s.stream);
    }
    if (rc) return rc;
    // one copy back: everything up to the end of the used part of `state`
    const size_t used = s.off_state + (size_t)s.num_nodes;
    if (!s.direct && hipMemcpyAsync(s.out_host, s.out_dev, used, hipMemcpyDeviceToHost, s.stream) != hipSuccess)
        return fail(DGCN_ERR_LAUNCH, "dgcn_host_solver: device-to-host copy failed");
    return DGCN_OK;
}

}  // namespace dgcn

using namespace dgcn;

extern "C" {

int dgcn_host_solver_create(const DgcnModel* model, const double* dinv_table, int32_t table_len, int32_t predict_mwis,
                            float x_const, int32_t want_scores, int32_t depth, int32_t pack_threads, DgcnHostSolver** out) {
    if ((model && (!model->layers_host || model->num_layers < 1 || !dinv_table || table_len < 1)) || !out || depth < 1 || depth > 64)
        return fail(DGCN_ERR_ARG, "dgcn_host_solver_create: bad argument");
    DgcnHostSolver* h = new DgcnHostSolver;
    h->lgs_only = model == nullptr;
    h->model = DgcnModel{};
    if (model) {
        h->layers.assign(model->layers_host, model->layers_host + model->num_layers);  // the descriptors are copied, the
        h->model = *model;                                                              // weights they point at are not
        h->model.layers_host = h->layers.data();
    }
    h->table = dinv_table;
    h->table_len = table_len;
    h->predict_mwis = predict_mwis;
    h->x_const = x_const;
    h->pack_threads = pack_threads;
    h->want_scores = (want_scores && model) ? 1 : 0;
    (void)hipGetDevice(&h->device);
    h->slots.resize(depth);
    if (hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) != hipSuccess) {
        dgcn_host_solver_destroy(h);
        return fail(DGCN_ERR_WORKSPACE, "dgcn_host_solver_create: cannot create a stream");
    }
    for (auto& s : h->slots) {
        void* cnt = nullptr;
        if (hipMalloc(&cnt, 256) == hipSuccess && hipMemset(cnt, 0, 256) == hipSuccess) s.done_count = static_cast<uint32_t*>(cnt);
        else if (cnt) (void)hipFree(cnt);
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.copied, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) {
            dgcn_host_solver_destroy(h);
            return fail(DGCN_ERR_WORKSPACE, "dgcn_host_solver_create: cannot create a stream / event");
        }
    }
    *out = h;
    return DGCN_OK;
}

void dgcn_host_solver_destroy(DgcnHostSolver* h) {
    if (!h) return;
    DeviceScope on_device(h->device);
    for (auto& s : h->slots) {
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        free_in(s);
        free_out(s);
        if (s.ws) (void)hipFree(s.ws);
        if (s.exp_dev) (void)hipFree(s.exp_dev);
        if (s.done_count) (void)hipFree(s.done_count);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.copied) (void)hipEventDestroy(s.copied);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    delete h;
}

int dgcn_host_solver_submit(DgcnHostSolver* h, const void* const* indptr_host, const void* const* indices_host,
                            const double* const* weights_host, const int32_t* num_nodes_host, int32_t num_graphs,
                            int32_t index_bytes) {
    if (!h || num_graphs < 0) return fail(DGCN_ERR_ARG, "dgcn_host_solver_submit: bad argument");
    DeviceScope on_device(h->device);
    const int k = h->next;
    DgcnHostSolver::Slot& s = h->slots[k];
    if (s.busy)
        return fail(DGCN_ERR_ARG, "dgcn_host_solver_submit: slot %d still holds an unread result (read it before submitting %zu more batches)",
                    k, h->slots.size());
    DgcnPackInfo info;
    int rc = dgcn_pack_measure(indptr_host, num_nodes_host, num_graphs, index_bytes, weights_host ? 1 : 0, &info, nullptr);
    if (rc) return rc;
    if ((rc = ensure_in(s, (size_t)info.total_bytes))) return rc;
    if (h->lgs_only && !weights_host) return fail(DGCN_ERR_ARG, "dgcn_host_solver_submit: the greedy search needs weights");
    int threads = h->pack_threads;
    if (threads <= 0) threads = 8;
    // A batch of up to a few dozen graphs: the kernel reads it where the packer left it and writes its results where the
    // caller reads them (pinned memory is device-addressable) - two copy commands and the gaps around them cost more
    // than PCIe round trips inside the kernel (tools/direct_probe.py: 101 vs 110 us for one N = 200 graph, 150 vs 170 us
    // for 63 of them = 1.1 MB; option "host_direct_bytes": largest batch handled this way, default 2 MB).
    const int64_t direct_opt = opt64(OPT_HOST_DIRECT_BYTES);
    const size_t direct_bytes = direct_opt >= 0 ? (size_t)direct_opt : (size_t)(2 << 20);
    const bool direct = (size_t)info.total_bytes <= direct_bytes;
    // Larger batches cross PCIe in the compact form (16-bit local column ids, 16-bit degrees: about half the bytes, and
    // half the bytes for the packer to write) and are expanded on the device (expand.hip); graphs beyond 65 535 vertices,
    // and every batch when option "host_compact" is 0, go as the ordinary block-diagonal CSR.
    const bool compact_ok = opt(OPT_HOST_COMPACT) != 0;
    DgcnCompactInfo ci = {};
    bool compact = false;
    if (!direct && compact_ok && info.num_nodes > 0 && compact_layout(&info, &ci) == 0) {
        rc = pack_compact(indptr_host, indices_host, weights_host, num_nodes_host, num_graphs, index_bytes, s.in_host, s.in_cap, &info, &ci,
                          threads, h->lgs_only);
        if (rc < 0) return rc;
        compact = rc == 0;
    }
    if (!compact) {
        rc = pack_batch(indptr_host, indices_host, weights_host, num_nodes_host, num_graphs, index_bytes, s.in_host, s.in_cap, &info,
                        threads, h->lgs_only);
        if (rc) return rc;
    }
    if (!h->lgs_only && info.max_degree >= h->table_len)
        return fail(DGCN_ERR_ARG, "dgcn_host_solver_submit: vertex degree %d beyond the d^-1/2 table (%d entries)", info.max_degree,
                    h->table_len);
    char* base = static_cast<char*>(direct ? s.in_host_dev : s.in_dev);
    DgcnBatch b;
    b.num_graphs = info.num_graphs;
    b.num_nodes = info.num_nodes;
    b.num_edges = info.num_edges;
    b.max_nodes = info.max_nodes;
    b.max_graph_edges = info.max_graph_edges;
    size_t exp_col_off = 0;
    // A compact batch for the deep-stack kernel stays compact: the kernel's image build reads degrees and 16-bit local
    // columns itself (no expansion launch, which - starved of CUs by the previous batch's solve - ended with that solve and
    // left 17 us between two solves; nor its 8.4 MB of row pointers and columns).  The one-layer kernel and the plain greedy
    // search take the expanded form.
    s.compact_direct = false;
    if (compact) {
        b.graph_ptr = reinterpret_cast<const int32_t*>(base + ci.off_graph_ptr);
        b.row_ptr = nullptr;
        b.col_idx = nullptr;
        const bool direct_ok = opt(OPT_HOST_COMPACT_DIRECT) != 0;  // (tests switch it)
        s.compact_direct = direct_ok && !h->lgs_only && dgcn_solve_path(&b, &h->model) == 1 && !shallow_takes(&b, &h->model);
        if (s.compact_direct) {
            s.c_edge = reinterpret_cast<const int32_t*>(base + ci.off_edge_ptr);
            s.c_deg = reinterpret_cast<const unsigned short*>(base + ci.off_deg);
            s.c_col = reinterpret_cast<const unsigned short*>(base + ci.off_col);
        } else {
            exp_col_off = align16(((size_t)info.num_nodes + 1) * 4);
            if ((rc = ensure_exp(s, exp_col_off + align16((size_t)std::max(info.num_edges, 1) * 4)))) return rc;
            b.row_ptr = reinterpret_cast<const int32_t*>(static_cast<char*>(s.exp_dev));
            b.col_idx = reinterpret_cast<const int32_t*>(static_cast<char*>(s.exp_dev) + exp_col_off);
        }
    } else {
        b.graph_ptr = reinterpret_cast<const int32_t*>(base + info.off_graph_ptr);
        b.row_ptr = reinterpret_cast<const int32_t*>(base + info.off_row_ptr);
        b.col_idx = reinterpret_cast<const int32_t*>(base + info.off_col_idx);
    }
    if (!h->lgs_only && !dgcn_solve_path(&b, &h->model))
        return fail(DGCN_ERR_UNSUPPORTED, "dgcn_host_solver_submit: this model / batch shape is outside the fused kernel and outside the any-size path");
    if ((rc = ensure_out(s, info.num_nodes, info.num_graphs, h->want_scores != 0))) return rc;
    if (!h->lgs_only) {
        const size_t need = dgcn_solve_workspace(&b, &h->model);
        if ((rc = ensure_ws(s, need))) return rc;
    }
    s.num_nodes = info.num_nodes;
    s.num_graphs = info.num_graphs;
    s.direct = direct;
    s.batch = b;
    s.off_weights = compact ? ci.off_weights : info.off_weights;
    s.done_total = 0;
    const size_t copy_bytes = compact ? (size_t)ci.total_bytes : (size_t)info.total_bytes;
    if (info.num_graphs > 0 && info.num_nodes > 0) {
        if (direct) {
            int32_t* st = reinterpret_cast<int32_t*>(static_cast<char*>(s.out_host) + s.off_status);
            st[0] = 0;
            // the latency path (one slot, nothing else to do meanwhile): a completion word written by the kernel itself
            const bool word_ok = opt(OPT_HOST_DONE_WORD) != 0;
            if (word_ok && h->slots.size() == 1 && !h->lgs_only && s.done_count && dgcn_solve_path(&b, &h->model) == 1) {
                s.done_base += (uint32_t)info.num_graphs;
                s.done_total = s.done_base;
                st[2] = 0;
            }
        } else {
            // one slot: nothing to overlap with, everything on the slot's stream (no cross-stream hop on the latency path);
            // several: copy - and the expansion of a compact batch - on the shared copy stream, beside the previous
            // batch's solve on its slot's stream, which then only waits for the `copied` event
            hipStream_t in_stream = h->slots.size() == 1 ? s.stream : h->copy_stream;
            if (hipMemcpyAsync(s.in_dev, s.in_host, copy_bytes, hipMemcpyHostToDevice, in_stream) != hipSuccess)
                return fail(DGCN_ERR_LAUNCH, "dgcn_host_solver_submit: host-to-device copy failed");
            if (compact && !s.compact_direct && (rc = expand_compact(s.in_dev, &ci, info.num_graphs, info.num_nodes, info.max_nodes,
                                                static_cast<int32_t*>(s.exp_dev), reinterpret_cast<int32_t*>(static_cast<char*>(s.exp_dev) + exp_col_off),
                                                in_stream)))
                return rc;
            if (in_stream != s.stream &&
                (hipEventRecord(s.copied, in_stream) != hipSuccess || hipStreamWaitEvent(s.stream, s.copied, 0) != hipSuccess))
                return fail(DGCN_ERR_LAUNCH, "dgcn_host_solver_submit: host-to-device copy failed");
        }
        if ((rc = launch_slot(h, s))) return rc;
    } else {  // nothing to launch: graphs without vertices have total 0 after 0 rounds
        std::memset(s.out_host, 0, s.off_state);
    }
    if (hipEventRecord(s.done, s.stream) != hipSuccess) return fail(DGCN_ERR_LAUNCH, "dgcn_host_solver_submit: hipEventRecord failed");
    s.busy = true;
    h->next = (k + 1) % (int)h->slots.size();
    return k;
}

int dgcn_host_solver_result(DgcnHostSolver* h, int32_t slot, const uint8_t** state, const double** totals,
                            const int32_t** rounds, const float** scores, int32_t* status_bits, int32_t* num_nodes,
                            int32_t* num_graphs) {
    if (!h || slot < 0 || slot >= (int)h->slots.size()) return fail(DGCN_ERR_ARG, "dgcn_host_solver_result: bad slot");
    DgcnHostSolver::Slot& s = h->slots[slot];
    if (!s.busy) return fail(DGCN_ERR_ARG, "dgcn_host_solver_result: slot %d holds no result", slot);
    DeviceScope on_device(h->device);
    bool seen = false;
    if (s.done_total) {  // the kernel's own word first; the event as well now and then (a fault path may not count itself)
        volatile const int32_t* word = reinterpret_cast<volatile const int32_t*>(static_cast<const char*>(s.out_host) + s.off_status + 8);
        for (unsigned spins = 1;; ++spins) {
            if ((uint32_t)*word == s.done_total) { seen = true; break; }
            if ((spins & 255u) == 0 && hipEventQuery(s.done) != hipErrorNotReady) break;
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!seen && hipEventSynchronize(s.done) != hipSuccess) return fail(DGCN_ERR_LAUNCH, "dgcn_host_solver_result: waiting for the batch failed");
    s.busy = false;
    const char* oh = static_cast<const char*>(s.out_host);
    int32_t bits = *reinterpret_cast<const int32_t*>(oh + s.off_status);
    if ((bits & DGCN_FAULT_CLUSTER) && !h->lgs_only && s.num_nodes > 0) {
        // The several-workgroups-per-graph variant of the fused kernel found its workgroups on different XCDs (a partition
        // mode with another dispatch order) or lost one: its results cannot be trusted.  Switch the variant off for the
        // rest of the process and solve the batch again - it is still where the packer put it.
        dgcn_set_cluster(0);
        if (s.direct) *reinterpret_cast<int32_t*>(static_cast<char*>(s.out_host) + s.off_status) = 0;
        else if (hipMemsetAsync(static_cast<char*>(s.out_dev) + s.off_status, 0, 4, s.stream) != hipSuccess)
            return fail(DGCN_ERR_LAUNCH, "dgcn_host_solver_result: clearing the status word failed");
        s.done_total = 0;  // (no completion word for the second run: the first one has already counted its graphs)
        int rc = launch_slot(h, s);
        if (rc) return rc;
        if (hipStreamSynchronize(s.stream) != hipSuccess) return fail(DGCN_ERR_LAUNCH, "dgcn_host_solver_result: waiting for the batch failed");
        bits = *reinterpret_cast<const int32_t*>(oh + s.off_status);
    }
    if (state) *state = reinterpret_cast<const uint8_t*>(oh + s.off_state);
    if (totals) *totals = reinterpret_cast<const double*>(oh + s.off_totals);
    if (rounds) *rounds = reinterpret_cast<const int32_t*>(oh + s.off_rounds);
    if (scores) *scores = h->want_scores ? reinterpret_cast<const float*>(oh + s.off_scores) : nullptr;
    if (status_bits) *status_bits = bits;
    if (num_nodes) *num_nodes = s.num_nodes;
    if (num_graphs) *num_graphs = s.num_graphs;
    if (bits && !s.direct) {  // the status word accumulates: clear it for the slot's next batch
        (void)hipMemsetAsync(static_cast<char*>(s.out_dev) + s.off_status, 0, 4, s.stream);
    }
    return DGCN_OK;
}

}  // extern "C"
