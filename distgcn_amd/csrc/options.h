// Process-wide switches of libdgcn.so behind dgcn_set_option / dgcn_get_option (include/dgcn.h "Options").
// One table, one atomic word per switch: no environment variable is read anywhere in the library, nothing is cached in a
// function-local static, a switch set by one thread is seen by the next call of every thread.  -1 means "automatic"
// wherever a switch chooses between code paths that produce the same results.
#pragma once
#include <cstdint>

namespace dgcn {

// X(enumerator, "key", default)
#define DGCN_OPTION_LIST(X)                                                                                          \
    X(OPT_FUSED_CLUSTER, "fused_cluster", -1)           /* workgroups per graph of the small-batch launch: -1 auto, 0 / 1 off, K >= 2 forced */ \
    X(OPT_FUSED_BLOCK, "fused_block", -1)               /* threads per workgroup of k_fused: -1 auto, 512, 1024 */    \
    X(OPT_FUSED_GW, "fused_gw", -1)                     /* k_fused with values and gather words in global scratch (two workgroups per CU for large images): -1 auto, 0 never, 1 wherever it fits */ \
    X(OPT_FUSED_ORDER, "fused_order", -1)               /* largest-graphs-first dispatch order: -1 auto, 0 off, 1 on */ \
    X(OPT_FUSED_FOLD, "fused_fold", -1)                 /* one-round launches with two workgroups per CU: the second half of the dispatch order smallest first: -1 auto, 0 off, N >= 1 forced at position N */ \
    X(OPT_NARROW_PAD, "narrow_pad", -1)                 /* deep stacks narrower than 32 beyond 512 vertices: -1 zero-padded onto k_big / k_big2 where that pays, 0 never, 1 wherever a kernel takes the copy */ \
    X(OPT_GENERAL, "general", -1)                       /* 1: every shape takes the any-size path; 0: never; -1 auto */ \
    X(OPT_SHALLOW, "shallow", -1)                       /* 0: one-layer models do not take k_shallow */              \
    X(OPT_SHALLOW_LONG, "shallow_long", -1)             /* k_shallow's long-row variant: -1 auto, 0 / 1 */            \
    X(OPT_SHALLOW_BLOCK, "shallow_block", -1)           /* k_shallow threads per workgroup: -1 auto, 64 .. 1024 */   \
    X(OPT_WIDE1, "wide1", -1)                           /* 0: one-layer models beyond 512 vertices run layer by layer */ \
    X(OPT_WIDE2, "wide2", -1)                           /* 0: two-layer models beyond 512 vertices run layer by layer */ \
    X(OPT_WIDE_AHEAD, "wide_ahead", -1)                 /* 0: the greedy rounds of k_wide1 / k_lgs / k_big2 do not use ahead lists */ \
    X(OPT_ROLLOUT_BITS, "rollout_bits", -1)             /* 0: rollout completions as instance launches, not an instance per bit */ \
    X(OPT_BIG, "big", -1)                               /* 0: no k_big / k_big2 (layer by layer) */                   \
    X(OPT_BIG_SOLVE, "big_solve", -1)                   /* 0: supports and greedy search in launches of their own */   \
    X(OPT_BIG_BLOCK, "big_block", -1)                   /* k_big threads per workgroup: -1 auto, 512, 1024 */         \
    X(OPT_BIG_TILES, "big_tiles", -1)                   /* k_big tiles per wave: -1 auto, 2, 4 */                     \
    X(OPT_BIG_RESIDUAL, "big_residual", -1)             /* 0: residual steps of deep models through the compaction launches */ \
    X(OPT_BIG2, "big2", -1)                             /* 1: every shape k_big takes goes to k_big2; 0: k_big2 off */ \
    X(OPT_TAIL, "tail", -1)                             /* 0: DGCN_RESIDUAL_FINISH_SMALL is ignored */                \
    X(OPT_LAYER_FUSE, "layer_fuse", -1)                 /* 0: mode 0 never fuses an aggregation with the next transform */ \
    X(OPT_LGS_LPV, "lgs_lpv", -1)                       /* k_lgs lanes per vertex: -1 auto, 1 .. 64 */                \
    X(OPT_LGS_BLOCK, "lgs_block", -1)                   /* k_lgs threads per workgroup: -1 auto, 512, 1024 */         \
    X(OPT_SPMM_PAD, "spmm_pad", 0)                      /* tuning of k_spmm_lds (tools/tune_spmm*.py) */              \
    X(OPT_SPMM_GLOBAL, "spmm_global", 0)                                                                              \
    X(OPT_SPMM_ROWS, "spmm_rows", 0)                                                                                  \
    X(OPT_SPMM_BLOCK, "spmm_block", 0)                                                                                \
    X(OPT_SPMM_CSRCAP, "spmm_csrcap", -1)                                                                             \
    X(OPT_SPMM_SPLIT, "spmm_split", 0)                                                                                \
    X(OPT_HOST_DIRECT_BYTES, "host_direct_bytes", -1)   /* host solver: largest batch the kernel reads from pinned memory; -1 = 2 MB */ \
    X(OPT_HOST_COMPACT, "host_compact", -1)             /* host solver: compact transfer form: -1 auto, 0 / 1 */      \
    X(OPT_HOST_COMPACT_DIRECT, "host_compact_direct", -1)                                                             \
    X(OPT_HOST_DONE_WORD, "host_done_word", -1)         /* 0: the host solver waits for the stream, not for the kernels' completion word */ \
    X(OPT_TEST_CLUSTER_FAULT, "test_cluster_fault", 0)  /* test hook: the cluster launch reports DGCN_FAULT_CLUSTER although there is none */ \
    X(OPT_DIAG_FLAGS, "diag_flags", 0)                  /* -DDGCN_DIAG builds only: ablation bits of the kernel in question */ \
    X(OPT_DIAG_STAMPS, "diag_stamps", 0)                /* -DDGCN_DIAG builds only: device address of the phase-clock array */

enum Opt : int {
#define DGCN_OPT_ENUM(e, k, d) e,
    DGCN_OPTION_LIST(DGCN_OPT_ENUM)
#undef DGCN_OPT_ENUM
    OPT_COUNT
};

int64_t opt64(Opt o);                                  // relaxed atomic load
inline int opt(Opt o) { return (int)opt64(o); }
void opt_store(Opt o, int64_t v);

}  // namespace dgcn
