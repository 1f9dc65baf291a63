// K1 instantiations: dot family (cosine, inner product, negative inner product).
#include "vt_scan.cuh"

#include <cstdlib>

namespace vt {
namespace dev {
hipError_t launch_scan_dot(const ScanDev &sd, uint32_t blocks, size_t lds, bool padded, hipStream_t s) {
  // A/B switch for tuning: lane order resolved at run time instead of compile time
  const bool rt_order = env::on(env::SCAN_RT_ORDER);
  if (rt_order && !padded && sd.a.k <= (uint32_t)kSmallK) return launch_scan_t<OP_DOT, -1, kCapSmall, false, false>(sd, blocks, lds, s);
  VT_SCAN_DISPATCH_ORDERED(OP_DOT);
}
}  // namespace dev
}  // namespace vt
