// K1 instantiations: manhattan, lane order at compile time like dot / L2.
// (Through round 2 this metric ran on one build with the order resolved at run time: the ISA of
// its hot loop had 980 instructions and 70 branches per group of eight loads where the dot
// build has 498 and none -- 5.59 ms against 4.58 ms per scan of 30.7 GB.)
#include "vt_scan.cuh"

namespace vt {
namespace dev {
hipError_t launch_scan_l1(const ScanDev &sd, uint32_t blocks, size_t lds, bool padded, hipStream_t s) {
  VT_SCAN_DISPATCH_ORDERED(OP_L1);
}
}  // namespace dev
}  // namespace vt
