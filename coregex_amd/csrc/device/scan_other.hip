// Placeholders until the char-class and Teddy kernels land (programs of those kinds are not marked
// supported yet, so these are unreachable).
#include <hip/hip_runtime.h>
#include "scan_dfa.h"
namespace cxgdev {
hipError_t launch_scan_teddy(const ScanArgs&, hipStream_t) { return hipErrorNotSupported; }
}  // namespace cxgdev
