// K1 instantiations: L2 family (l2, l2_squared).
#include "vt_scan.cuh"

namespace vt {
namespace dev {
hipError_t launch_scan_l2(const ScanDev &sd, uint32_t blocks, size_t lds, bool padded, hipStream_t s) {
  VT_SCAN_DISPATCH_ORDERED(OP_L2);
}
}  // namespace dev
}  // namespace vt
