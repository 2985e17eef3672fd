// capi_state.hip — error state, launch-mode state per device, per-thread scratch: the definitions behind capi_internal.hpp.
#include "capi_internal.hpp"

namespace cxgapi {

thread_local std::string t_err;
thread_local int t_device = 0;
std::atomic<bool> g_exiting{false};
PathState g_path[16];
thread_local ScratchSet t_scratch_set;
thread_local bool t_u32Rows = false;
thread_local Scratch::AsyncSlot* t_asyncSlot = nullptr;

int deviceCount() {
  static int n = -1;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (n >= 0) return n;
  int c = 0;
  std::atexit([] { g_exiting.store(true); });
  if (hipGetDeviceCount(&c) != hipSuccess) { (void)hipGetLastError(); c = 0; }
  int ok = 0;
  for (int d = 0; d < c; d++) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
  }
  n = (ok == c) ? c : 0;   // only an all-gfx950 box is accepted
  return n;
}

// Wait for the call's stream.  (Round 5 had a CXG_SPIN_SYNC knob that polled hipStreamQuery in front of this: 4 us saved per call, and rows
// of a relaunch ladder's last kernel read before they were written — profiles/r05_pytest_gpu_spin_sync.log.  The cause was never
// found — hipStreamSynchronize followed the poll unconditionally, so an early "ready" cannot explain it — and the knob
// is gone.)
hipError_t syncStream(hipStream_t stream) { return hipStreamSynchronize(stream); }

int getScratch(Scratch** out) {
  if (deviceCount() <= 0) return fail(CXG_E_NO_GPU, "no gfx950 device visible (this library has no CPU search path)");
  if (t_device < 0 || t_device >= deviceCount() || t_device >= 16) return fail(CXG_E_INVALID, "bad device index");
  HIP_TRY(hipSetDevice(t_device));
  Scratch& s = t_scratch[t_device];
  if (s.device < 0) {
    HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    for (auto& e : s.ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.hostCtl), 64, hipHostMallocDefault));
    s.device = t_device;
  }
  *out = &s;
  return CXG_OK;
}

int ensureStatus(Scratch& s, uint64_t ntiles) {
  if (ntiles <= s.statusCap && s.ctl) return CXG_OK;
  if (s.ctl) HIP_TRY(hipFree(s.ctl));
  s.ctl = nullptr; s.status = nullptr; s.statusCap = 0;
  uint64_t cap = ntiles + ntiles / 4 + 1024;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.ctl), 64 + 2 * cap * sizeof(uint64_t)));   // look-back words, then the exit-state words of scan_fsm.hip
  s.status = reinterpret_cast<uint64_t*>(s.ctl + 64);
  s.statusCap = cap;
  s.needZero = true;
  return CXG_OK;
}

int deviceCopy(const std::vector<uint8_t>& host, void** slot, const uint8_t** out) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (!*slot) {   // device copies are a cache, the program stays logically immutable
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, host.size()));
    HIP_TRY(hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
    *slot = d;
  }
  *out = static_cast<const uint8_t*>(*slot);
  return CXG_OK;
}
int deviceBlob(const cxg_program* p, int device, const uint8_t** out) {
  return deviceCopy(p->blob, &const_cast<cxg_program*>(p)->dev[device], out);
}

// CXG_DIGIT_KERNEL=1|2 force the first (nested-loop) / second (flat) table-walking generation (A/B profiling);
// default 6 = bit-parallel chain kernel (scan_chain_wave.hip; also serves UseDFA programs that are one chain) when
// the program allows it, else generation 2; a tile that raises the fallback flag hands the scan to generation 2
// (UseDFA: the bidirectional table kernel).  Generations 3-5 (candidate list, workgroup chain, wave prefilter) were
// stepping stones of round 1 and are gone (git history).
int digitKernelGeneration() {
  static const int g = [] { const char* e = getenv("CXG_DIGIT_KERNEL"); const int v = e ? atoi(e) : 6; return (v == 1 || v == 2) ? v : 6; }();
  return g;
}

uint64_t tilesFor(uint32_t kind, uint64_t len) {
  (void)kind;
  return (len + cxgdev::kTile - 1) / cxgdev::kTile;
}


}  // namespace cxgapi
