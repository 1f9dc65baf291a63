// K1 instantiations for gathered rows (rerank of candidate lists, stateless
// vector_top_k batches): operation and lane order at run time.
#include "vt_scan.cuh"

namespace vt {
namespace dev {
hipError_t launch_scan_general(const ScanDev &sd, uint32_t blocks, uint32_t nq, size_t lds, hipStream_t s) {
  if (sd.a.k <= (uint32_t)kSmallK) return launch_scan_t<-1, -1, kCapSmall, true, true>(sd, blocks, lds, s, nq);
  return launch_scan_t<-1, -1, kCapLarge, true, true>(sd, blocks, lds, s, nq);
}
}  // namespace dev
}  // namespace vt
