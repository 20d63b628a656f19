// Fused per-graph persistent forward (mode 1).  Placeholder until the kernel lands: the entry
// points exist so the dispatcher links, and report "unsupported" loudly.
#include "common.h"

namespace dgcn {

size_t fused_workspace(const DgcnBatch*, const DgcnModel*) { return 256; }

int fused_forward(const DgcnBatch*, const DgcnCsr*, const DgcnModel*, const float*, float, float*, void*, size_t,
                  hipStream_t) {
    return fail(DGCN_ERR_UNSUPPORTED, "dgcn_gcn_forward_batch: fused mode is not built in this version");
}

}  // namespace dgcn
