// K1 instantiations for gathered rows (rerank of candidate lists, stateless
// vector_top_k batches) and for queries too long for LDS: operation and lane order at run time.
// The two operations a K2b batch re-scores with (dot, squared L2) in the default lane order have
// builds of their own in vt_scan_gather.hip.
#include "vt_scan.cuh"

namespace vt {
namespace dev {
hipError_t launch_scan_general(const ScanDev &sd, uint32_t blocks, uint32_t nq, size_t lds, hipStream_t s) {
  const bool small = sd.a.k <= (uint32_t)kSmallK;
  if (sd.p.q_global) {
    if (small) return launch_scan_t<-1, -1, kCapSmall, true, true, true>(sd, blocks, lds, s, nq);
    return launch_scan_t<-1, -1, kCapLarge, true, true, true>(sd, blocks, lds, s, nq);
  }
  if (sd.a.order == kDefaultReduceOrder) {
    const int op = metric_op(sd.a.metric);
    if (op == OP_DOT || op == OP_L2) return launch_scan_gather(sd, blocks, nq, lds, s);
  }
  if (small) return launch_scan_t<-1, -1, kCapSmall, true, true>(sd, blocks, lds, s, nq);
  return launch_scan_t<-1, -1, kCapLarge, true, true>(sd, blocks, lds, s, nq);
}
}  // namespace dev
}  // namespace vt
