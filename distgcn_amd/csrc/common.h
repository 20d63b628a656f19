// Shared host-side helpers for libdgcn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/dgcn.h"
#include "options.h"

namespace dgcn {

constexpr int kWave = 64;        // CDNA wavefront width
constexpr float kLeakyAlpha = 0.2f;

// thread-local error text behind dgcn_last_error()
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

// kernel-family timing (dgcn_timing_*): RAII pair of events around one launch
// With timing enabled a launch goes through hipExtLaunchKernelGGL, which stamps the slot's two events with
// the kernel's own begin / end (the duration rocprofv3 reports), not with the gaps around it.
struct TimedLaunch {
    TimedLaunch(const char* family, hipStream_t stream);
    bool timed() const { return slot >= 0; }
    hipEvent_t start_ev() const;
    hipEvent_t stop_ev() const;
    int slot;
    int device;  // whose slot table the pair belongs to (runtime.hip)
    hipStream_t stream;
};

#define DGCN_LAUNCH(t, kernel, grid, block, lds, stream, ...)                                                  \
    do {                                                                                                       \
        if ((t).timed())                                                                                       \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, (t).start_ev(), (t).stop_ev(), 0, __VA_ARGS__); \
        else                                                                                                   \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                 \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(DGCN_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return DGCN_OK;
}

__device__ __forceinline__ float apply_act(float x, int act) {
    if (act == DGCN_ACT_LEAKY_RELU) return x > 0.0f ? x : kLeakyAlpha * x;
    if (act == DGCN_ACT_RELU) return x > 0.0f ? x : 0.0f;
    return x;
}

// Completion word for the one-graph latency path (host_solver.hip -> fused.hip): the next dgcn_solve_batch of this thread
// makes every graph's finishing workgroup count itself in `count` (device memory) once its outputs have left the CU, and the
// one that reaches `target` stores it to `flag` (pinned host memory) - the host can stop waiting some microseconds before
// the queue's own end-of-kernel signal arrives.  Consumed (cleared) by that call.
struct DoneHook {
    int32_t* flag = nullptr;
    uint32_t* count = nullptr;
    uint32_t target = 0;
};
extern thread_local DoneHook g_done_hook;

// The batch of the next dgcn_solve_batch of this thread is in the compact transfer form (include/dgcn.h: DgcnCompactInfo):
// b->row_ptr / b->col_idx are null, the fused kernel takes degrees + 16-bit local columns straight into its image build
// (host_solver.hip: no expansion launch between the copy and the solve).  Consumed (cleared) by that call; only set for
// (batch, model) pairs that go to the deep-stack kernel (not the one-layer kernel, not the plain greedy search).
struct CompactHook {
    const int32_t* edge_ptr = nullptr;       // [num_graphs + 1] first entry of every graph
    const unsigned short* deg = nullptr;     // [num_nodes] entries per row
    const unsigned short* col = nullptr;     // [num_edges] column ids, local to their graph
};
extern thread_local CompactHook g_compact_hook;
bool shallow_takes(const DgcnBatch* b, const DgcnModel* m);  // shallow.hip: one-layer models go to k_shallow

// dgcn_pack_batch with one more check for callers whose kernel cannot report it (pack.hip)
int pack_batch(const void* const* indptr_host, const void* const* indices_host, const double* const* weights_host,
               const int32_t* num_nodes_host, int32_t num_graphs, int32_t index_bytes, void* staging_host,
               size_t staging_bytes, DgcnPackInfo* info, int32_t num_threads, bool reject_self_loops);

// the compact transfer format (include/dgcn.h: DgcnCompactInfo; pack.hip: pack_compact, expand.hip: expand_compact)
int compact_layout(const DgcnPackInfo* std_info, DgcnCompactInfo* ci);  // 0, or 1 = this batch cannot be compacted
int pack_compact(const void* const* indptr_host, const void* const* indices_host, const double* const* weights_host,
                 const int32_t* num_nodes_host, int32_t num_graphs, int32_t index_bytes, void* staging_host, size_t staging_bytes,
                 DgcnPackInfo* info, const DgcnCompactInfo* ci, int32_t num_threads, bool reject_self_loops);  // DGCN_OK, < 0, or 1 = not compactable
int expand_compact(const void* compact_dev, const DgcnCompactInfo* ci, int32_t num_graphs, int32_t num_nodes, int32_t max_nodes,
                   int32_t* row_ptr_out, int32_t* col_idx_out, hipStream_t stream);

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace dgcn
