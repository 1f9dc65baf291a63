// K1 instantiations: chebyshev, float hamming and jaccard (order-insensitive: max / exact
// small-integer sums); manhattan has its own file (vt_scan_l1.hip).
#include "vt_scan.cuh"

namespace vt {
namespace dev {

#define VT_SCAN_DISPATCH_FIXED(OPV, ORD)                                                      \
  do {                                                                                        \
    const bool big = sd.a.k > (uint32_t)kSmallK;                                                             \
    if (!padded) {                                                                            \
      if (!big) return launch_scan_t<OPV, ORD, kCapSmall, false, false>(sd, blocks, lds, s);          \
      return launch_scan_t<OPV, ORD, kCapLarge, false, false>(sd, blocks, lds, s);                    \
    }                                                                                         \
    if (!big) return launch_scan_t<OPV, ORD, kCapSmall, false, true>(sd, blocks, lds, s);             \
    return launch_scan_t<OPV, ORD, kCapLarge, false, true>(sd, blocks, lds, s);                       \
  } while (0)

hipError_t launch_scan_misc(const ScanDev &sd, uint32_t blocks, size_t lds, bool padded, hipStream_t s) {
  switch (metric_op(sd.a.metric)) {
    case OP_L1: return launch_scan_l1(sd, blocks, lds, padded, s);
    case OP_LINF: VT_SCAN_DISPATCH_FIXED(OP_LINF, 0);
    case OP_HAM: VT_SCAN_DISPATCH_FIXED(OP_HAM, 0);
    default: VT_SCAN_DISPATCH_FIXED(OP_JAC, 0);
  }
}
}  // namespace dev
}  // namespace vt
