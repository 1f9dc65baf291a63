// K1 instantiations: dot family (cosine, inner product, negative inner product).
#include "vt_scan.cuh"

#include <cstdlib>

namespace vt {
namespace dev {
hipError_t launch_scan_dot(const ScanDev &sd, uint32_t blocks, size_t lds, bool padded, hipStream_t s) {
  VT_SCAN_DISPATCH_ORDERED(OP_DOT);
}
}  // namespace dev
}  // namespace vt
