// K1 instantiations for gathered rows with the operation and the lane order fixed at compile time:
// dot and squared L2 in the default order -- what the exact re-scoring of a K2b batch's candidates
// and the cosine / dot / L2 reranks run.  Everything else goes through vt_scan_general.hip.
#include "vt_scan.cuh"

namespace vt {
namespace dev {
hipError_t launch_scan_gather(const ScanDev &sd, uint32_t blocks, uint32_t nq, size_t lds, hipStream_t s) {
  const bool small = sd.a.k <= (uint32_t)kSmallK;
  if (metric_op(sd.a.metric) == OP_DOT) {
    if (small) return launch_scan_t<OP_DOT, kDefaultReduceOrder, kCapSmall, true, true>(sd, blocks, lds, s, nq);
    return launch_scan_t<OP_DOT, kDefaultReduceOrder, kCapLarge, true, true>(sd, blocks, lds, s, nq);
  }
  if (small) return launch_scan_t<OP_L2, kDefaultReduceOrder, kCapSmall, true, true>(sd, blocks, lds, s, nq);
  return launch_scan_t<OP_L2, kDefaultReduceOrder, kCapLarge, true, true>(sd, blocks, lds, s, nq);
}
}  // namespace dev
}  // namespace vt
