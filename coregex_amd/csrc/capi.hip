// capi.hip — the extern "C" surface of libcoregex_hip.so (include/coregex_hip.h).
// Host glue only: program construction (host/), device copies, per-thread stream + scratch
// (the SearchState analogue, meta/search_state.go:23-62), launches (device/).  There is no CPU
// search path in this library: without a gfx950 device every search entry returns CXG_E_NO_GPU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/coregex_hip.h"
#include "device/scan_dfa.h"
#include "device/synth.hpp"
#include "device/block_common.hpp"
#include "device/walk.hpp"
#include "device/wave_common.hpp"
#include "device/fsm.hpp"
#include "device/bt.hpp"
#include "host/frontend.h"
#include "host/program.h"

namespace cxgdev {
hipError_t launch_scan_dfa(uint32_t kind, const ScanArgs& a, uint32_t fwd_states, uint32_t rev_states, hipStream_t stream);
size_t scan_dfa_dynamic_lds(uint32_t fwd_states, uint32_t rev_states);
hipError_t launch_scan_charclass(const ScanArgs& a, hipStream_t stream);
hipError_t launch_scan_digit_flat(const ScanArgs& a, uint32_t fwd_states, hipStream_t stream);
hipError_t launch_scan_chain_wave(const ScanArgs& a, uint32_t ncls, bool sets, bool caps, hipStream_t stream);
hipError_t launch_scan_fields_wave(const ScanArgs& a, hipStream_t stream, bool* persistent);   // scan_fields_wave.hip
hipError_t launch_scan_delim_wave(const ScanArgs& a, hipStream_t stream);   // scan_delim_wave.hip
int fields_shape(const ChainAux& c);
int literal_shape(const ChainAux& c);
int trio_shape(const ChainAux& c);
hipError_t launch_scan_trio_wave(const ScanArgs& a, hipStream_t stream, bool* persistent);   // scan_fields_wave.hip
hipError_t launch_scan_teddy(const ScanArgs& a, hipStream_t stream);
hipError_t launch_scan_teddy_wave(const ScanArgs& a, uint32_t verify_dfa_states, hipStream_t stream);
hipError_t launch_scan_charclass_wave(const ScanArgs& a, hipStream_t stream);
hipError_t launch_scan_fsm(const ScanArgs& a, uint32_t lds_bytes, bool shallow, int look, hipStream_t stream, uint32_t direct_bytes = 0, bool lean = false);
}  // namespace cxgdev

namespace {

thread_local std::string t_err;
thread_local int t_device = 0;

int fail(int code, const std::string& msg) { t_err = msg; return code; }
int failHip(hipError_t e, const char* what) {
  t_err = std::string(what) + ": " + hipGetErrorString(e);
  return CXG_E_DEVICE;
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) return failHip(_e, #expr);    \
  } while (0)

std::atomic<bool> g_exiting{false};   // set by an atexit hook: the HIP runtime may already be gone, leave its memory to the OS

// Fast-path state per device, process-wide.  Three launch modes rest on how the device dispatches workgroups — static group
// assignment (workgroups arrive in index order), the persistent fields kernel (its whole grid is co-resident) and the delimiter
// kernel (index order) — and each has a spin watchdog that turns a broken assumption into an error bit instead of a hang.  Another
// tenant of the GPU can break them for a while, so a watchdog hit is a DEMOTION WITH A TERM, not a verdict (round 4 latched
// "never again" for the whole process): the call that was hit reruns one mode down, the next `penalty` calls that would have
// used the mode stay one mode down, then the mode is tried again; a second hit doubles the term (8, 16, ... 1024 calls), a clean
// call on the mode resets it.  cxg_path_state() shows the counters to the host.
struct PathMode {
  std::atomic<uint32_t> penalty{0};     // calls left one mode down
  std::atomic<uint32_t> backoff{8};     // term of the next demotion
  std::atomic<uint32_t> hits{0};        // watchdog hits since the process started
  bool allowed() const { return penalty.load(std::memory_order_relaxed) == 0; }
  void consume() {                      // a call that wanted the mode and was kept off it
    uint32_t v = penalty.load(std::memory_order_relaxed);
    while (v != 0 && !penalty.compare_exchange_weak(v, v - 1, std::memory_order_relaxed)) {}
  }
  void demote() {
    hits.fetch_add(1, std::memory_order_relaxed);
    const uint32_t b = backoff.load(std::memory_order_relaxed);
    penalty.store(b, std::memory_order_relaxed);
    backoff.store(b >= 512 ? 1024 : b * 2, std::memory_order_relaxed);
  }
  void clean() { backoff.store(8, std::memory_order_relaxed); }
};
struct PathState {
  PathMode staticGroups, persistent, delim;
  // ONE launch section at a time per device from THIS process (every goroutine of a cgo host may be scanning): two persistent grids
  // would each hold half the CUs and wait for waves that cannot become resident, and a persistent grid beside a static-group kernel
  // waits just the same (measured in round 5: two threads scanning 1 GiB each ran into the 0.4 s watchdog); the scans are HBM-bound,
  // so callers lose nothing by taking turns.  Round 5 held a mutex from launch to completion — and, for asynchronous calls, until
  // cxg_wait: a handle that was never waited for blocked every other thread.  Now the turns are taken ON THE
  // DEVICE: a launch section (OrderGate below) makes its stream wait for the completion event of the section in front of it,
  // enqueues its kernels, and records its own completion event; the mutex only guards that event while the section is being
  // enqueued (microseconds), nothing is held across a synchronisation or an API boundary.
  std::mutex orderMutex;
  hipEvent_t orderEvent = nullptr;      // completion of the last launch section any thread enqueued on this device
  bool orderValid = false;
  std::atomic<uint32_t> orderWaiters{0};
};
struct OrderGate {
  PathState& ps;
  hipStream_t stream;
  std::unique_lock<std::mutex> lk;
  OrderGate(PathState& p, hipStream_t st) : ps(p), stream(st), lk(p.orderMutex, std::defer_lock) {
    ps.orderWaiters.fetch_add(1, std::memory_order_relaxed); lk.lock(); ps.orderWaiters.fetch_sub(1, std::memory_order_relaxed);
    if (!ps.orderEvent && hipEventCreateWithFlags(&ps.orderEvent, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ps.orderEvent = nullptr; }
    if (ps.orderEvent && ps.orderValid && hipStreamWaitEvent(stream, ps.orderEvent, 0) != hipSuccess) (void)hipGetLastError();
  }
  void close() {                        // everything of the section is enqueued: the next section (any thread) runs behind it
    if (!lk.owns_lock()) return;
    if (ps.orderEvent) { if (hipEventRecord(ps.orderEvent, stream) == hipSuccess) ps.orderValid = true; else (void)hipGetLastError(); }
    lk.unlock();
  }
  ~OrderGate() { close(); }
  OrderGate(const OrderGate&) = delete;
  OrderGate& operator=(const OrderGate&) = delete;
};
PathState g_path[16];

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

// Per-thread scratch for one in-flight call per device.
struct Scratch {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  uint8_t* ctl = nullptr;        // ticket(4) pad(4) total(8) err(4) pad(4) ... 8 XCD tickets at +32 -> 64 B, in front of `status`
  uint64_t* status = nullptr;    // ctl + 64: one allocation, one memset per launch
  uint64_t statusCap = 0;
  uint64_t* fsmMaps = nullptr;   // scan_fsm.hip: three map words per group
  uint64_t fsmMapsCap = 0;
  uint64_t* hostCtl = nullptr;   // pinned mirror of ctl
  uint32_t epoch = 0;            // last launch epoch used on `status` (block_common.hpp kEpochShift), 1..1023
  bool needZero = true;          // the next epoch launch must start from a zeroed control block + status array
  uint64_t* prof = nullptr;      // CXG_PROF phase counters (device)
  static constexpr size_t kProfRecords = 1u << 18;
  uint8_t* hay = nullptr; uint64_t hayCap = 0;     // staging for host haystacks
  int64_t* findRow = nullptr;                       // cxg_find_device: the one row (16 bytes used)
  int64_t* out = nullptr; uint64_t outCap = 0;     // staging for host result arrays (rows*width)
  uint8_t* pinHay = nullptr;     // small host haystacks: pinned, read by the kernels over PCIe (no copy calls)
  int64_t* pinOut = nullptr;     // ... and their rows, written straight into pinned host memory
  uint8_t* bt = nullptr; size_t btCap = 0;         // k_captures_bt: per-thread visited bitmap + stack
  uint32_t* pfStatus = nullptr; uint64_t pfCap = 0; uint32_t pfEpoch = 0;   // k_scan_fields_pers: one word per unit, own 16-bit launch epoch
  uint32_t* pfTickets = nullptr;   // ... [32][64] ticket counters a cache line apart, one block per launch epoch (scan_fields_wave.hip, round 6)
  uint64_t* pfRec = nullptr; uint64_t pfRecRounds = 0;   // ... 128 records of 16 bytes per round, tagged with the same epoch
  uint64_t* pfStats = nullptr;                           // ... per wave: units that waited, polls (CXG_VERBOSE)
  int64_t* offSpans = nullptr; uint64_t offSpansCap = 0;  // offset captures (scanOffsetCaps): the spans in front of the expansion kernel
  int64_t* nullRows = nullptr; uint64_t nullRowsCap = 0;  // nullable programs (scanNullable): rows of the non-empty variant,
  uint64_t* nullCov = nullptr; uint64_t nullCovCap = 0;   // ... inclusive sums of the positions they cover, + one sum per block of 4096 rows
  uint8_t* bothHay = nullptr; uint64_t bothHayCap = 0;    // UseBoth restart (scanDevice): aligned copy of the haystack's suffix
  int64_t* bothRows = nullptr; uint64_t bothRowsCap = 0;  // ... rows of a launch whose caller gave no room for them
  unsigned long long* bothFirst = nullptr;                // ... index of the first row longer than the restart span
  // cxg_find_all_device_async: launches of this thread that have not been waited for yet (ring of kAsyncSlots)
  struct AsyncSlot {
    bool busy = false, done = false;                // done: the call ran synchronously (a program without an async-capable first launch)
    int done_rc = 0; uint64_t done_n = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    uint64_t* ctl = nullptr;                        // two pinned words: total, error
    const cxg_program* p = nullptr; const void* hay = nullptr; uint64_t len = 0; int64_t base = 0, limit = 0; void* out = nullptr; uint64_t cap = 0; void* user_stream = nullptr;
    hipStream_t stream = nullptr;
    uint32_t kernelId = 0, mode = 0;                // mode: 1 static groups, 2 persistent, 3 delimiter kernel (what a clean finish resets)
    uint64_t tiles = 0;
    cxg_timing timing;
  };
  static constexpr int kAsyncSlots = 16;
  AsyncSlot async[kAsyncSlots];
  uint64_t* asyncCtl = nullptr;                     // pinned, 2 words per slot
  int asyncInFlight = 0;
  // Everything above belongs to ONE OS thread.  A cgo host moves goroutines across many threads, so the scratch is
  // released when its thread exits (thread_local destructor) or on request (cxg_thread_release).
  void release() {
    if (device < 0) return;
    if (hipSetDevice(device) == hipSuccess) {
      if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
      for (auto& e : ev) if (e) (void)hipEventDestroy(e);
      if (ctl) (void)hipFree(ctl);
      if (fsmMaps) (void)hipFree(fsmMaps);
      if (pfStatus) (void)hipFree(pfStatus);
      if (pfRec) (void)hipFree(pfRec);
      if (pfTickets) (void)hipFree(pfTickets);
      if (findRow) (void)hipFree(findRow);
      if (pfStats) (void)hipFree(pfStats);
      if (prof) (void)hipFree(prof);
      if (hay) (void)hipFree(hay);
      if (out) (void)hipFree(out);
      if (bt) (void)hipFree(bt);
      if (offSpans) (void)hipFree(offSpans);
      if (nullRows) (void)hipFree(nullRows);
      if (nullCov) (void)hipFree(nullCov);
      if (bothHay) (void)hipFree(bothHay);
      if (bothRows) (void)hipFree(bothRows);
      if (bothFirst) (void)hipFree(bothFirst);
      if (hostCtl) (void)hipHostFree(hostCtl);
      if (asyncCtl) (void)hipHostFree(asyncCtl);
      for (auto& as : async) for (auto& e : as.ev) if (e) (void)hipEventDestroy(e);
      if (pinHay) (void)hipHostFree(pinHay);
      if (pinOut) (void)hipHostFree(pinOut);
    }
    (void)hipGetLastError();
    *this = Scratch();
  }
};
struct ScratchSet {
  Scratch s[16];
  ~ScratchSet() { if (!g_exiting.load()) for (auto& x : s) x.release(); }
};
thread_local ScratchSet t_scratch_set;
#define t_scratch t_scratch_set.s
// Staging buffers above this size are returned after the call instead of being kept for the thread's lifetime.
constexpr uint64_t kKeepStagingBytes = 256ull << 20;

// Wait for the call's stream.  CXG_SPIN_SYNC=1 polls hipStreamQuery for up to 2 ms before parking the thread (hipStreamSynchronize is
// woken ~10 us after the kernel ended: 0.2419 -> 0.2352 ms per 1 GiB call) — OFF by default: with it on, the device fuzz and
// tests/test_gpu_parity.py::test_random_patterns of round 5 returned rows the last kernel of a relaunch ladder had not written yet
// (profiles/r05_pytest_gpu_spin_sync.log: the tail of the array still held an earlier call's rows) — hipStreamQuery answered "ready"
// before the stream had drained.  Correctness first; the knob stays for measurements.
hipError_t syncStream(hipStream_t stream) {
  static const bool spin = getenv("CXG_SPIN_SYNC") != nullptr;
  if (spin) {
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t i = 0;; i++) {
      const hipError_t q = hipStreamQuery(stream);
      if (q == hipSuccess) break;
      if (q != hipErrorNotReady) return q;
      if ((i & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
  }
  return hipStreamSynchronize(stream);
}

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

// (The small kernels outside scanDeviceOnce — merges of nullable programs, capture expansions, corpus fills — are launch sections too:
// OrderGate gate(g_path[device], stream) around their launches.)

uint64_t tilesFor(uint32_t kind, uint64_t len);
int scanNullable(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out, uint64_t cap,
                 uint64_t* n_out, void* user_stream, cxg_timing* timing);
int scanOffsetCaps(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out, uint64_t cap,
                   uint64_t* n_out, void* user_stream, cxg_timing* timing, int row_width);
int scanNullableSubmatch(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out, uint64_t cap,
                         uint64_t* n_out, void* user_stream, cxg_timing* timing, int row_width);
thread_local bool t_u32Rows = false;
thread_local Scratch::AsyncSlot* t_asyncSlot = nullptr;            // cxg_find_all_device_async in progress on this thread: leave the first launch pending if it can be
constexpr int kRcPending = -1001;                                  // (internal) scanDeviceOnce left its launch in the slot                               // cxg_find_all_device_u32 in progress on this thread (ScanArgs::u32_rows)

// CXG_DIGIT_KERNEL=1|2 force the first (nested-loop) / second (flat) table-walking generation (A/B profiling);
// default 6 = bit-parallel chain kernel (scan_chain_wave.hip; also serves UseDFA programs that are one chain) when
// the program allows it, else generation 2; a tile that raises the fallback flag hands the scan to generation 2
// (UseDFA: the bidirectional table kernel).  Generations 3-5 (candidate list, workgroup chain, wave prefilter) were
// stepping stones of round 1 and are gone (git history).
int digitKernelGeneration() {
  static const int g = [] { const char* e = getenv("CXG_DIGIT_KERNEL"); const int v = e ? atoi(e) : 6; return (v == 1 || v == 2) ? v : 6; }();
  return g;
}

__global__ void k_captures(const uint8_t* hay, int64_t hay_base, int64_t* rows, uint64_t nrows, uint32_t width,
                           const uint8_t* capblob, uint32_t* err) {
  const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
  if (i >= nrows) return;
  const cxgdev::CapHeader* ch = reinterpret_cast<const cxgdev::CapHeader*>(capblob);
  cxgdev::CapView cv{capblob + ch->next_off, capblob + ch->maskid_off, capblob + ch->fin_off,
                     reinterpret_cast<const uint32_t*>(capblob + ch->masks_off), ch->n_entries, ch->start_entry};
  // rows hold absolute offsets (hay_base added); the walk indexes the device buffer, so shift the pointer
  if (!cxgdev::capture_walk(cv, hay - hay_base, rows + i * width, width)) cxgdev::raise_err(err, 4u);
}

// Capture pass, fast form: the one-pass table (next | maskid << 8 per entry and byte) staged in LDS (dynamic size),
// the first 64+ bytes of every match fetched with five 16-byte loads issued together (one memory latency per
// match instead of one per 4 bytes) and parked in the thread's LDS slot, the slots of a row kept in registers as
// offsets from the match start and written once (one 64-byte row per thread for three groups).
// MAXS = slots held in registers; wider rows and bigger tables use k_captures.
constexpr uint32_t kCapLdsEntries = 48;
constexpr int kCapSlotDwords = 21;                                  // 80 bytes + 1 dword of bank skew per thread
template <int MAXS>
__global__ __launch_bounds__(256) void k_captures_lds(const uint8_t* hay, int64_t hay_base, uint64_t hay_len, int64_t* rows, uint64_t nrows,
                                                      uint32_t width, const uint8_t* capblob, uint32_t* err) {
  extern __shared__ __attribute__((aligned(16))) uint16_t s_tab[];  // [n_entries][256]
  __shared__ uint32_t s_hay[256 * kCapSlotDwords];
  __shared__ uint32_t s_masks[256];
  __shared__ uint8_t s_fin[kCapLdsEntries];
  const cxgdev::CapHeader* ch = reinterpret_cast<const cxgdev::CapHeader*>(capblob);
  const uint32_t ne = ch->n_entries;
  const uint8_t* gnext = capblob + ch->next_off;
  const uint8_t* gmid = capblob + ch->maskid_off;
  for (uint32_t i = threadIdx.x; i < ne * 256u; i += blockDim.x) s_tab[i] = static_cast<uint16_t>(gnext[i] | (gmid[i] << 8));
  for (uint32_t i = threadIdx.x; i < ch->n_masks && i < 256u; i += blockDim.x) s_masks[i] = reinterpret_cast<const uint32_t*>(capblob + ch->masks_off)[i];
  for (uint32_t i = threadIdx.x; i < ne; i += blockDim.x) s_fin[i] = capblob[ch->fin_off + i];
  __syncthreads();
  const uint8_t* h0 = hay - hay_base;                              // rows hold absolute offsets (hay_base added)
  const uint64_t lim16 = (reinterpret_cast<uint64_t>(hay) + hay_len + 15u) & ~15ull;   // 16-byte loads stay below this
  uint32_t* slot = s_hay + threadIdx.x * kCapSlotDwords;
  const uint8_t* slotb = reinterpret_cast<const uint8_t*>(slot);
  bool bad = false;
  for (uint64_t r = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; r < nrows; r += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    int64_t* row = rows + r * width;
    const longlong2 se = *reinterpret_cast<const longlong2*>(row);
    const int64_t s = se.x, e = se.y;
    const uint64_t a0 = (reinterpret_cast<uint64_t>(h0) + static_cast<uint64_t>(s)) & ~15ull;
    const uint32_t skew = static_cast<uint32_t>((reinterpret_cast<uint64_t>(h0) + static_cast<uint64_t>(s)) & 15u);
    uint4 q[5];
#pragma unroll
    for (int k = 0; k < 5; k++) q[k] = (a0 + 16u * k + 16u <= lim16) ? *reinterpret_cast<const uint4*>(a0 + 16u * k) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 5; k++) { slot[4 * k] = q[k].x; slot[4 * k + 1] = q[k].y; slot[4 * k + 2] = q[k].z; slot[4 * k + 3] = q[k].w; }
    // 32-bit loop state (a match is at most window + serial-walk budget long); the staged part and the rare rest of a
    // long match are separate loops so that the hot one has no global-memory branch
    const int32_t len = static_cast<int32_t>(e - s);
    const int32_t nst = len < 80 - static_cast<int32_t>(skew) ? len : 80 - static_cast<int32_t>(skew);
    int32_t v[MAXS];
#pragma unroll
    for (int k = 0; k < MAXS; k++) v[k] = -1;
    uint32_t ent = ch->start_entry;
    int32_t i = 0;
    auto step = [&](uint32_t b) -> bool {
      const uint32_t t = s_tab[ent * 256u + b];
      const uint32_t nx = t & 0xFFu;
      if (nx == 0xFFu) return false;
      const uint32_t m = s_masks[t >> 8];
      if (m) {
#pragma unroll
        for (int k = 2; k < MAXS; k++) if ((m >> k) & 1u) v[k] = i;
      }
      ent = nx;
      return true;
    };
    const uint8_t* sb = slotb + skew;
    for (; i < nst; i++) if (!step(sb[i])) { bad = true; break; }
    if (!bad) {
      const uint8_t* gb = h0 + s;
      for (; i < len; i++) if (!step(gb[i])) { bad = true; break; }
    }
    const uint32_t f = s_fin[ent];
    if (f == 0xFFu) bad = true;
    else {
      const uint32_t m = s_masks[f];
#pragma unroll
      for (int k = 2; k < MAXS; k++) if ((m >> k) & 1u) v[k] = len;
    }
#pragma unroll
    for (int k = 2; k + 1 < MAXS; k += 2) {
      if (static_cast<uint32_t>(k) < width) {
        longlong2 o;
        o.x = v[k] < 0 ? -1 : s + v[k];
        o.y = v[k + 1] < 0 ? -1 : s + v[k + 1];
        *reinterpret_cast<longlong2*>(row + k) = o;
      }
    }
  }
  if (bad) cxgdev::raise_err(err, 4u);
}

// Capture pass, general form (patterns that are not one-pass): bounded backtracking over the NFA per match row
// (device/bt.hpp).  One thread per row, grid-stride; every thread owns 16 KiB of scratch in HBM (visited bitmap + stack).
// Two tiers: k_captures_bt_lds first (256 threads per workgroup, 256 bytes of LDS scratch per thread, the NFA image in LDS when
// it fits: as many resident threads as the CUs hold), rows it cannot finish are marked and redone by k_captures_bt.
template <bool LOOK>
__global__ __launch_bounds__(256) void k_captures_bt_lds(const uint8_t* hay, int64_t hay_base, uint64_t hay_len, int64_t* rows, uint64_t nrows, uint32_t width,
                                                         const uint8_t* btblob, uint32_t img_lds_bytes, uint32_t* err) {
  extern __shared__ __attribute__((aligned(16))) uint8_t s_bt[];   // [img_lds_bytes] image, then per-thread scratch
  __shared__ uint64_t s_stack[256 * cxgdev::kBtSmallStack];
  __shared__ uint32_t s_vis[256 * cxgdev::kBtSmallVisited];
  const cxgdev::BtHeader* gh = reinterpret_cast<const cxgdev::BtHeader*>(btblob);
  for (uint32_t i = threadIdx.x; i < img_lds_bytes / 4u; i += blockDim.x) reinterpret_cast<uint32_t*>(s_bt)[i] = reinterpret_cast<const uint32_t*>(btblob)[i];
  __syncthreads();
  const cxgdev::BtHeader* h = img_lds_bytes ? reinterpret_cast<const cxgdev::BtHeader*>(s_bt) : gh;
  uint64_t* stack = s_stack + threadIdx.x * cxgdev::kBtSmallStack;
  uint32_t* visited = s_vis + threadIdx.x * cxgdev::kBtSmallVisited;
  uint32_t bad = 0;
  for (uint64_t r = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; r < nrows; r += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    int64_t* row = rows + r * width;
#pragma unroll
    for (uint32_t i = 0; i < cxgdev::kBtSmallVisited; i++) visited[i] = 0u;
    const uint32_t rc = cxgdev::bt_captures<LOOK>(h, hay - hay_base, row, width, visited, stack, cxgdev::kBtSmallVisited, cxgdev::kBtSmallStack,
                                            hay_base, hay_base + static_cast<int64_t>(hay_len));   // (bounds: read by assertion states only)
    if (rc == 1u) row[2] = cxgdev::kBtRowPending;                 // left to the large tier (its slots are rewritten there)
    else bad |= rc;
  }
  if (bad & 2u) cxgdev::raise_err(err, 4u);
}

template <bool LOOK>
__global__ __launch_bounds__(64) void k_captures_bt(const uint8_t* hay, int64_t hay_base, uint64_t hay_len, int64_t* rows, uint64_t nrows, uint32_t width,
                                                    const uint8_t* btblob, uint8_t* scratch, uint32_t* err) {
  const cxgdev::BtHeader* h = reinterpret_cast<const cxgdev::BtHeader*>(btblob);
  const uint64_t tid = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
  uint32_t* visited = reinterpret_cast<uint32_t*>(scratch + tid * (cxgdev::kBtVisitedWords * 4ull + cxgdev::kBtStackEntries * 8ull));
  uint64_t* stack = reinterpret_cast<uint64_t*>(visited + cxgdev::kBtVisitedWords);
  uint32_t bad = 0;
  for (uint64_t r = tid; r < nrows; r += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    int64_t* row = rows + r * width;
    if (row[2] != cxgdev::kBtRowPending) continue;                 // the small tier finished this row
    const uint64_t bits = (static_cast<uint64_t>(row[1] - row[0]) + 1) * h->n_states;
    const uint32_t nw = bits > static_cast<uint64_t>(cxgdev::kBtVisitedWords) * 32u ? 0u : static_cast<uint32_t>((bits + 31) >> 5);
    for (uint32_t i = 0; i < nw; i++) visited[i] = 0u;
    bad |= cxgdev::bt_captures<LOOK>(h, hay - hay_base, row, width, visited, stack, cxgdev::kBtVisitedWords, cxgdev::kBtStackEntries,
                               hay_base, hay_base + static_cast<int64_t>(hay_len));   // rows hold absolute offsets (hay_base added)
  }
  if (bad & 1u) cxgdev::raise_err(err, cxgdev::kErrSerialLimit);   // a match too long for the per-row budget: this haystack is left to the caller
  if (bad & 2u) cxgdev::raise_err(err, 4u);
}

// One resident round of capture workgroups: as many as the LDS footprint lets a CU hold (grid-stride over the rows),
// so every workgroup stages the table once and all finish together.
unsigned captureGrid(uint64_t nrows, uint32_t dyn_lds) {
  const uint32_t lds = 256u * kCapSlotDwords * 4u + 1024u + 64u + dyn_lds;
  uint32_t per_cu = (160u * 1024u) / lds;
  if (per_cu > 8u) per_cu = 8u;
  if (per_cu < 1u) per_cu = 1u;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return static_cast<unsigned>(std::min<uint64_t>((nrows + 255) / 256, static_cast<uint64_t>(cus) * per_cu));
}

// Capture pass of FindAllSubmatch: one thread per match row behind the span kernel on the same stream — the one-pass table from LDS
// or HBM, or bounded backtracking per row (device/bt.hpp) for patterns that are not one-pass.  The row count is only known on the
// device, so it is read back first (one 8-byte copy).
int launchCapturePass(const cxg_program* p, Scratch& s, const cxgdev::ScanArgs& a, const uint8_t* d_cap, hipStream_t stream, uint32_t& launches) {
  if (!a.epoch) HIP_TRY(hipMemcpyAsync(s.hostCtl, s.ctl, 32, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  uint64_t nrows = s.hostCtl[1];
  if (nrows > a.cap) nrows = a.cap;
  if (static_cast<uint32_t>(s.hostCtl[2]) & (8u | 2u)) nrows = 0;   // the span kernel asked for a rerun: its rows are not final
  if (nrows) {
    const cxgdev::CapHeader* chh = reinterpret_cast<const cxgdev::CapHeader*>(p->capBlob.data());
    const bool lds_ok = chh->magic != cxgdev::kBtMagic && chh->n_entries <= kCapLdsEntries && chh->n_masks <= 256u;
    if (chh->magic == cxgdev::kBtMagic) {                        // not one-pass: backtracking per row
      const unsigned blk = 64, grd = static_cast<unsigned>(std::min<uint64_t>((nrows + blk - 1) / blk, 64));   // <= 4096 threads x 16 KiB
      const size_t need = static_cast<size_t>(grd) * blk * (cxgdev::kBtVisitedWords * 4ull + cxgdev::kBtStackEntries * 8ull);
      if (s.btCap < need) {
        if (s.bt) HIP_TRY(hipFree(s.bt));
        s.bt = nullptr; s.btCap = 0;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.bt), need));
        s.btCap = need;
      }
      {
        const uint32_t img = reinterpret_cast<const cxgdev::BtHeader*>(p->capBlob.data())->total_bytes;
        const uint32_t img_lds = img <= 16384u ? ((img + 3u) & ~3u) : 0u;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const unsigned g1 = static_cast<unsigned>(std::min<uint64_t>((nrows + 255) / 256, static_cast<uint64_t>(cus) * 2u));
        if (p->capHasLook) hipLaunchKernelGGL(k_captures_bt_lds<true>, dim3(g1), dim3(256), img_lds, stream, a.hay, a.base, a.len, a.out, nrows, a.row_width, d_cap, img_lds, a.err);
        else hipLaunchKernelGGL(k_captures_bt_lds<false>, dim3(g1), dim3(256), img_lds, stream, a.hay, a.base, a.len, a.out, nrows, a.row_width, d_cap, img_lds, a.err);
      }
      // (patterns without assertions run the instantiation without the assertion branch: the walk of round 2's device runs)
      if (p->capHasLook) hipLaunchKernelGGL(k_captures_bt<true>, dim3(grd), dim3(blk), 0, stream, a.hay, a.base, a.len, a.out, nrows, a.row_width, d_cap, s.bt, a.err);
      else hipLaunchKernelGGL(k_captures_bt<false>, dim3(grd), dim3(blk), 0, stream, a.hay, a.base, a.len, a.out, nrows, a.row_width, d_cap, s.bt, a.err);
    } else if (lds_ok && a.row_width <= 8) {
      const unsigned grd = captureGrid(nrows, chh->n_entries * 512u);
      hipLaunchKernelGGL(k_captures_lds<8>, dim3(grd), dim3(256), chh->n_entries * 512u, stream, a.hay, a.base, a.len, a.out, nrows, a.row_width, d_cap, a.err);
    } else if (lds_ok && a.row_width <= 16) {
      const unsigned grd = captureGrid(nrows, chh->n_entries * 512u);
      hipLaunchKernelGGL(k_captures_lds<16>, dim3(grd), dim3(256), chh->n_entries * 512u, stream, a.hay, a.base, a.len, a.out, nrows, a.row_width, d_cap, a.err);
    } else {
      const unsigned blk = 128, grd = static_cast<unsigned>((nrows + blk - 1) / blk);
      hipLaunchKernelGGL(k_captures, dim3(grd), dim3(blk), 0, stream, a.hay, a.base, a.out, nrows, a.row_width, d_cap, a.err);
    }
    HIP_TRY(hipGetLastError());
    launches = 2;
  }
    return CXG_OK;
}

// scanDeviceOnce: one search from the haystack's first byte.  kRcLongMatch (internal): a UseBoth program met a match longer
// than its restart span; *n_out = rows of plain leftmost-first iteration, the rows themselves are in d_out when it has room.
constexpr int kRcLongMatch = -1000;
int scanDeviceOnce(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out,
                   uint64_t cap, uint64_t* n_out, void* user_stream, cxg_timing* timing, int row_width) {
  if (!p) return fail(CXG_E_INVALID, "null program");
  const bool submatch = row_width > 2;
  if (submatch) {
    if (!p->subSupported) return fail(CXG_E_UNSUPPORTED, p->subWhyNot.empty() ? "submatch unsupported for this program" : p->subWhyNot);
  } else if (!p->supported) return fail(CXG_E_UNSUPPORTED, p->whyNot.empty() ? "unsupported program" : p->whyNot);
  if (n_out) *n_out = 0;
  if (timing) std::memset(timing, 0, sizeof *timing);
  if (limit == 0) return CXG_OK;  // Count(n == 0) == 0, meta/findall.go:298
  Scratch* sp;
  if (int rc = getScratch(&sp)) return rc;
  Scratch& s = *sp;
  if (len == 0) return CXG_OK;    // non-nullable patterns never match the empty haystack
  if (reinterpret_cast<uintptr_t>(d_hay) & 15u) return fail(CXG_E_INVALID, "device haystack must be 16-byte aligned");
  if (d_out && (reinterpret_cast<uintptr_t>(d_out) & 15u)) return fail(CXG_E_INVALID, "device output must be 16-byte aligned");
  const cxgdev::BlobHeader* h = reinterpret_cast<const cxgdev::BlobHeader*>(submatch ? p->subBlob.data() : p->blob.data());
  hipStream_t stream = user_stream ? static_cast<hipStream_t>(user_stream) : s.stream;
  const uint8_t* d_blob;
  const uint8_t* d_cap = nullptr;
  if (submatch) {
    cxg_program* mp = const_cast<cxg_program*>(p);
    if (int rc = deviceCopy(p->subBlob, &mp->devSub[t_device], &d_blob)) return rc;
    if (int rc = deviceCopy(p->capBlob, &mp->devCap[t_device], &d_cap)) return rc;
  } else if (int rc = deviceBlob(p, t_device, &d_blob)) return rc;
  // general-DFA kernel (scan_fsm.hip): first choice for programs the bit-parallel / literal kernels do not take, and
  // the fallback of those kernels (match-dense input, input without synchronising bytes)
  static const bool fsmOk = getenv("CXG_NO_FSM") == nullptr;
  const std::vector<uint8_t>& fsmImg = submatch ? p->subFsmBlob : p->fsmBlob;
  const uint8_t* d_fsm = nullptr;
  if (fsmOk && !fsmImg.empty()) {
    cxg_program* mp = const_cast<cxg_program*>(p);
    if (int rc = deviceCopy(fsmImg, submatch ? &mp->devSubFsm[t_device] : &mp->devFsm[t_device], &d_fsm)) return rc;
  }
  bool fsmTried = false;
  uint32_t lastReason = 0;
  cxgdev::ScanArgs a;
  a.pf_status = nullptr; a.pf_ticket = nullptr; a.pf_ncounters = 0;   // (set per launch by the fields programs' branch below)
  std::memset(&a.plan, 0, sizeof a.plan); a.plan_shape = 0;
  a.cc_nr = a.cc_neg = a.cc_pairs = 0; std::memset(a.cc_lo, 0, 4); std::memset(a.cc_hi, 0, 4);
  a.u32_rows = t_u32Rows ? 1u : 0u;
  if (a.u32_rows && (len >> 32) != 0) return fail(CXG_E_INVALID, "compact rows: the haystack must be shorter than 4 GiB (shard it)");
  a.hay = static_cast<const uint8_t*>(d_hay);
  a.len = len;
  a.base = base;
  a.blob = d_blob;
  a.out = static_cast<int64_t*>(d_out);
  a.cap = d_out ? cap : 0;
  if (limit > 0 && static_cast<uint64_t>(limit) < a.cap) a.cap = static_cast<uint64_t>(limit);
  a.row_width = static_cast<uint32_t>(row_width);
  a.ntiles = tilesFor(h->kind, len);
  if (a.ntiles > 0x7FFFFFFFull) return fail(CXG_E_INVALID, "haystack too large for one launch; shard it");
  {
    // look-back / exit words are indexed by GROUP, and the smallest group any kernel mode uses is the transducer kernel's mode 2:
    // one wave-tile per wave = 15 KiB, i.e. 1.07 groups per 16 KiB tile.  (Round 3 fix: a cached allocation that covered
    // `ntiles` of this call but not its mode-2 groups was written past its end — found by the CXG_NO_EPOCH run of the no-sync test.)
    const uint64_t smallest = static_cast<uint64_t>(cxgdev::kWaveTile) * cxgdev::kWavesPerBlock;
    const uint64_t maxGroups = (len + smallest - 1) / smallest + 1;
    if (int rc = ensureStatus(s, a.ntiles > maxGroups ? a.ntiles : maxGroups)) return rc;
  }
  a.status = s.status;
  a.status2 = s.status + s.statusCap;
  a.ticket = reinterpret_cast<uint32_t*>(s.ctl + 32);   // 8 per-XCD counters (block_common.hpp claim_tile)
  a.total = reinterpret_cast<uint64_t*>(s.ctl + 8);
  a.err = reinterpret_cast<uint32_t*>(s.ctl + 16);
  static const bool profOn = getenv("CXG_PROF") != nullptr;
  static const uint32_t dbgBits = getenv("CXG_DEBUG") ? static_cast<uint32_t>(atoi(getenv("CXG_DEBUG"))) : 0u;
  a.prof = nullptr;
  a.dbg = dbgBits;
  a.limit = limit > 0 ? static_cast<uint64_t>(limit) : 0;
  a.stop = reinterpret_cast<uint32_t*>(s.ctl + 24);                   // device word of the control block (zeroed with it; epoch-tagged otherwise)
  a.max_len = (h->flags & cxgdev::kFlagBothRestart) ? cxgdev::kBothRestartSpan : 0u;
  if (profOn) {
    // 16 summed counters, then one record of 8 timestamps per workgroup for the kernels that keep them (k_scan_charclass_wave)
    if (!s.prof) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.prof), 128 + Scratch::kProfRecords * 64));
    HIP_TRY(hipMemsetAsync(s.prof, 0, 128 + Scratch::kProfRecords * 64, stream));
    a.prof = s.prof;
  }
  int gen = digitKernelGeneration();
  uint32_t relaunches = 0;
  if (gen == 6 && !(h->flags & cxgdev::kFlagChainOrdered)) gen = 2;   // not a complete ordered chain: table-walking kernels
  if (h->kind == cxgdev::kKindDigit) {
    // gen stays 1, 2 or 6
  } else if (h->kind == cxgdev::kKindTeddy) {
    static const bool oldTeddy = getenv("CXG_TEDDY_KERNEL") && atoi(getenv("CXG_TEDDY_KERNEL")) == 1;
    gen = (oldTeddy || h->aux_len > 2048u) ? 0 : 7;                 // the wave kernel stages at most 2 KiB of literal tables
    // literals between assertions (walk.hpp TeddyAux::looks): the table kernel knows no assertions — wave kernel, else the transducer
    if (gen == 0 && reinterpret_cast<const cxgdev::TeddyAux*>(p->blob.data() + h->aux_off)->looks != 0u && !submatch) {
      if (!d_fsm) return fail(CXG_E_UNSUPPORTED, "literals between assertions: neither the wave kernel nor the transducer can take this program");
      gen = 10; fsmTried = true;
    }
    // 7 = wave kernel (scan_teddy_wave.hip), 0 = scan_teddy.hip
  } else if (h->kind == cxgdev::kKindCharClass) {
    static const bool oldCc = getenv("CXG_CC_KERNEL") && atoi(getenv("CXG_CC_KERNEL")) == 1;
    gen = (!oldCc && (h->flags & cxgdev::kFlagCcRanges)) ? 8 : 0;   // 8 = wave kernel (scan_charclass_wave.hip), 0 = scan_charclass.hip
    if ((h->flags & cxgdev::kFlagCcRanges) && reinterpret_cast<const cxgdev::CharClassAux*>(p->blob.data() + h->aux_off)->pairs) gen = 8;   // (the table kernel knows runs only)
  } else if (gen != 6) gen = 0;                                   // table kernels of the other kinds
  if (gen == 0 && h->kind == cxgdev::kKindBidir && (h->flags & cxgdev::kFlagPrefixLiteral)) {
    static const bool noPrefix = getenv("CXG_NO_PREFIX_KERNEL") != nullptr;
    if (!noPrefix) gen = 9;                                        // literal occurrences + anchored DFA walk (scan_teddy_wave.hip VERIFY)
  }
  if (d_fsm && (gen == 0 || (gen == 2 && digitKernelGeneration() == 6)) && (h->kind == cxgdev::kKindBidir || h->kind == cxgdev::kKindDigit)) {
    gen = 10;                                                       // table-walking kernels only when the transducer is unavailable or gives up
    fsmTried = true;
  }
  // `O [^E]+ E` programs: the delimiter kernel first (spans, no FindAll n: its kind look-back has no early stop), the transducer behind it
  static const bool delimOk = getenv("CXG_NO_DELIM_KERNEL") == nullptr;
  static const bool ticketsForced = getenv("CXG_TICKETS") != nullptr;
  PathState& ps = g_path[t_device];
  bool staticDenied = ticketsForced, persDenied = false;            // this call: a watchdog hit (or the environment) took the mode away
  if (delimOk && !submatch && gen == 10 && p->delim[3] != 0u && limit <= 0 && d_fsm && !ticketsForced) {
    if (ps.delim.allowed() && ps.staticGroups.allowed()) { gen = 11; fsmTried = false; }
    else ps.delim.consume();
  }
  if (h->kind == cxgdev::kKindFsmOnly) {                            // UseNFA programs (word boundaries): the transducer kernel is the only one
    if (!d_fsm) return fail(CXG_E_UNSUPPORTED, "program runs on the transducer kernel only (CXG_NO_FSM is set)");
    gen = 10;
    fsmTried = true;
  }
  // Wave kernels: static group assignment unless a look-back watchdog demoted it for a while (PathState above, block_common.hpp).
  if (gen >= 6 && !ticketsForced && !ps.staticGroups.allowed()) { staticDenied = true; ps.staticGroups.consume(); }
  static const bool fuseCapsOk = getenv("CXG_NO_FUSED_CAPTURES") == nullptr;
  bool fusedCaps = false;                                          // captures written by the chain kernel itself
  bool fieldsKernel = false;                                       // gen 6 served by scan_fields_wave.hip
  bool trioKernel = false;                                         // gen 6 served by k_scan_trio_wave
  bool persKernel = false;                                         // ... by k_scan_fields_pers (the launcher says)
  bool litKernel = false;                                          // ... by its literal mode (round 5)
  bool denseChain = p->denseChain[submatch ? 1 : 0].load(std::memory_order_relaxed) != 0;   // wave kernels: match-dense input seen before
  int fsmMode = p->fsmMode[submatch ? 1 : 0].load(std::memory_order_relaxed);                // transducer kernel: 0, 1 (dense), 2 (very dense)
  static const bool fsmDirectOk = getenv("CXG_FSM_NO_DIRECT") == nullptr;                     // A/B: the class-indexed tables for every machine
  static const bool fsmLeanOk = getenv("CXG_FSM_NO_LEAN") == nullptr;                         // A/B: k_scan_fsm for every machine
  // the lean kernel (scan_fsm.hip k_scan_fsml: shallow machines, entry states that collapse; byte-indexed rows where the image has them)
  // while it serves the program's input
  bool fsmDirect = fsmLeanOk && p->fsmNoDirect[submatch ? 1 : 0].load(std::memory_order_relaxed) == 0;
  bool fsmDirectRan = false, fsmDirectTables = false;
  uint8_t ladder[sizeof(cxg_timing{}.ladder)] = {0};               // kernel id of every span launch of this call, in order
  uint32_t nladder = 0;
  // One iteration = one span launch (+ its capture pass).  What comes next is decided at the bottom from the kernel's error word:
  // done; the same family in a denser mode; the transducer; the table-walking kernels — each `continue` below is one rung.
  for (;;) {
  fusedCaps = false;
  fieldsKernel = false;
  persKernel = false;
  litKernel = false;
  trioKernel = false;
  std::memset(a.caps, 0, sizeof a.caps);
  a.static_groups = (gen >= 6 && !staticDenied) ? 1u : 0u;
  OrderGate orderGate(ps, stream);                                 // this iteration's launch section: behind whatever any thread enqueued on the device before (closed once everything is enqueued)
  Scratch::AsyncSlot* const as = (t_asyncSlot && relaunches == 0 && !submatch && !profOn && !dbgBits && a.max_len == 0) ? t_asyncSlot : nullptr;
  if (gen == 11 && !a.static_groups) { gen = 10; fsmTried = true; }   // the delimiter kernel has no ticket mode
  a.ngroups = a.ntiles;
  if (gen == 8 || gen == 11) a.ngroups = (len + cxgdev::kCcGroupBytes - 1) / cxgdev::kCcGroupBytes;
  if (gen == 6 || gen == 7 || gen == 9 || gen == 10) a.ngroups = (len + cxgdev::kWaveGroupBytes - 1) / cxgdev::kWaveGroupBytes;
  a.tiles_per_wave = cxgdev::kTilesPerWave;
  if (((gen == 6 || gen == 7 || gen == 9) && denseChain) || (gen == 10 && fsmMode != 0)) {   // four times the row-buffer room per wave-tile
    a.tiles_per_wave = (gen == 10 && fsmMode == 2) ? 1u : static_cast<uint32_t>(cxgdev::kDenseTilesPerWave);   // transducer kernel, mode 2: one tile, 2048 rows
    const uint64_t gb = static_cast<uint64_t>(cxgdev::kWaveTile) * cxgdev::kWavesPerBlock * a.tiles_per_wave;
    a.ngroups = (len + gb - 1) / gb;
  }
  if (a.ngroups > s.statusCap) return fail(CXG_E_INTERNAL, "status words: more groups than the allocation covers");
  // Wave kernels with static groups tag their look-back words with a launch epoch and clear the next launch's error
  // word themselves: no memset between launches.  Everything else starts from a zeroed control block + status words.
  static const bool epochsOk = getenv("CXG_NO_EPOCH") == nullptr;
  const bool useEpoch = epochsOk && a.static_groups != 0;
  a.epoch = 0;
  a.total = reinterpret_cast<uint64_t*>(s.ctl + 8);
  a.err = reinterpret_cast<uint32_t*>(s.ctl + 16);
  if (gen == 10) {                                                 // three map words per group (scan_fsm.hip fsm_group_entry), epoch-tagged like the rest
    if (3 * a.ngroups > s.fsmMapsCap) {
      if (s.fsmMaps) HIP_TRY(hipFree(s.fsmMaps));
      s.fsmMaps = nullptr; s.fsmMapsCap = 0;
      const uint64_t cap = 3 * a.ngroups + 3 * a.ngroups / 4 + 1024;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.fsmMaps), cap * sizeof(uint64_t)));
      s.fsmMapsCap = cap;
      HIP_TRY(hipMemsetAsync(s.fsmMaps, 0, cap * sizeof(uint64_t), stream));
    }
    a.fsm_maps = s.fsmMaps;
  }
  // events only for a caller that asked for timing (the cgo shim does not); the "total" event only in front of a memset
  const bool wantEv = timing != nullptr && !(as != nullptr && useEpoch);
  const bool ev0 = wantEv && (!useEpoch || s.needZero || s.epoch >= 1023u);
  if (ev0) HIP_TRY(hipEventRecord(s.ev[0], stream));
  if (useEpoch) {
    if (s.needZero || s.epoch >= 1023u) {
      HIP_TRY(hipMemsetAsync(s.ctl, 0, 64 + 2 * s.statusCap * sizeof(uint64_t), stream));
      if (s.fsmMaps) HIP_TRY(hipMemsetAsync(s.fsmMaps, 0, s.fsmMapsCap * sizeof(uint64_t), stream));
      s.epoch = 0; s.needZero = false;
    }
    a.epoch = ++s.epoch;
    // total and error word in pinned host memory: written by the kernel (one store / a rare system-scope OR),
    // visible when the stream has drained, read here without a device-to-host copy
    s.hostCtl[1] = 0; s.hostCtl[2] = 0;
    a.total = s.hostCtl + 1;
    a.err = reinterpret_cast<uint32_t*>(s.hostCtl + 2);
    if (as) { as->ctl[0] = 0; as->ctl[1] = 0; a.total = as->ctl; a.err = reinterpret_cast<uint32_t*>(as->ctl + 1); }
  } else {
    // control block and the look-back words this launch will use, in one memset
    HIP_TRY(hipMemsetAsync(s.ctl, 0, 64 + a.ngroups * sizeof(uint64_t), stream));   // every kernel indexes status by group < ngroups <= ntiles
    if (gen == 11) HIP_TRY(hipMemsetAsync(a.status2, 0, a.ngroups * sizeof(uint64_t), stream));
    if (gen == 10) {
      HIP_TRY(hipMemsetAsync(a.status2, 0, a.ngroups * sizeof(uint64_t), stream));
      HIP_TRY(hipMemsetAsync(a.fsm_maps, 0, 3 * a.ngroups * sizeof(uint64_t), stream));
    }
    s.needZero = true;                                              // legacy words and error bits are left behind
  }
  const bool goAsync = as != nullptr && useEpoch;
  static const bool asyncTiming = getenv("CXG_ASYNC_TIMING") != nullptr;   // a start event per pending launch (cxg_wait's kernel_ms); off: one event per launch
  if (!goAsync) { if (wantEv) HIP_TRY(hipEventRecord(s.ev[1], stream)); }
  else if (asyncTiming) HIP_TRY(hipEventRecord(as->ev[0], stream));
  hipError_t le;
  a.blob = gen == 10 ? d_fsm : d_blob;
  if (a.u32_rows && a.out != nullptr && gen != 8 && gen != 6 && gen != 11)       // (gen 6: checked below, the persistent fields kernel only)
    return fail(relaunches ? CXG_E_INPUT : CXG_E_UNSUPPORTED, "compact rows (cxg_find_all_device_u32): this program's span kernel writes int64 rows only");
  if (gen == 10) {
    static const bool deepOnly = getenv("CXG_FSM_DEEP") != nullptr;   // A/B: the general event-list instantiation for every machine
    const cxgdev::FsmHeader* fh = reinterpret_cast<const cxgdev::FsmHeader*>(fsmImg.data());
    fsmDirectRan = fsmDirect && !deepOnly && fh->depth <= 1 && a.prof == nullptr && a.dbg == 0;
    fsmDirectTables = fsmDirectRan && fsmDirectOk && fh->direct_off != 0u && fh->nk == 1;
    le = cxgdev::launch_scan_fsm(a, fh->lds_bytes, fh->depth <= 1 && !deepOnly, fh->end_col != 0u ? 2 : (fh->nk > 1 ? 1 : 0), stream, fsmDirectTables ? fh->direct_bytes : 0u, fsmDirectRan);
  }
  else if (gen == 11) {
    std::memcpy(a.chain, p->delim, sizeof p->delim);
    le = cxgdev::launch_scan_delim_wave(a, stream);
  }
  else if (gen == 8) {
    // `Q[^Q]*Q` programs count EVENTS (occurrences of Q, two per row) in the look-back: FindAll's n is 2 n events
    const bool pairsProg = reinterpret_cast<const cxgdev::CharClassAux*>(p->blob.data() + h->aux_off)->pairs != 0u;
    cxgdev::ScanArgs b = a;
    {
      const cxgdev::CharClassAux* cax = reinterpret_cast<const cxgdev::CharClassAux*>(p->blob.data() + h->aux_off);
      static const bool plansOk = getenv("CXG_NO_PLANS") == nullptr;   // A/B: the generic range tests
      b.plan = cxgdev::plan_class(cax->nr, cax->lo, cax->hi);
      b.plan_shape = plansOk ? static_cast<uint32_t>(cxgdev::plan_shape(b.plan)) : 0u;
      b.cc_nr = cax->nr; b.cc_neg = cax->neg; b.cc_pairs = cax->pairs;
      for (int q = 0; q < 4; q++) { b.cc_lo[q] = cax->lo[q]; b.cc_hi[q] = cax->hi[q]; }
    }
    if (pairsProg) b.limit = a.limit * 2u;
    le = cxgdev::launch_scan_charclass_wave(b, stream);
  }
  else if (gen == 7) le = cxgdev::launch_scan_teddy_wave(a, 0, stream);
  else if (gen == 9) {                                              // required literal prefix + anchored DFA (kFlagPrefixLiteral)
    const uint8_t* hb = submatch ? p->subBlob.data() : p->blob.data();
    le = cxgdev::launch_scan_teddy_wave(a, reinterpret_cast<const cxgdev::TeddyAux*>(hb + h->aux_off)->dfa_states, stream);
  }
  else if (gen == 6) {
    const uint8_t* hb = submatch ? p->subBlob.data() : p->blob.data();
    std::memcpy(a.chain, hb + h->aux_off + 256, sizeof(cxgdev::ChainAux));
    if (!submatch && (h->flags & cxgdev::kFlagChainBounded)) std::memcpy(a.caps, p->chainBounds, sizeof a.caps);   // BND instantiation
    if (submatch && a.out && fuseCapsOk && p->chainCaps[0] && p->chainCaps[1] == a.row_width) {   // ChainCaps.on / .nslots
      std::memcpy(a.caps, p->chainCaps, sizeof a.caps);
      fusedCaps = true;
    }
    // fields programs (one field class, one separator class: the headline `\d+\.\d+\.\d+\.\d+`): the forward-only kernel;
    // match-dense input (a row buffer overflowed before) stays on the chain kernel's dense mode
    static const bool fieldsOk = getenv("CXG_NO_FIELDS_KERNEL") == nullptr;
    const bool fieldsCould = !(h->flags & (cxgdev::kFlagChainBounded | cxgdev::kFlagChainSets)) &&
                             cxgdev::fields_shape(*reinterpret_cast<const cxgdev::ChainAux*>(a.chain)) != 0;
    fieldsKernel = fieldsOk && !submatch && !denseChain && fieldsCould;
    // border-free literals over <= 4 distinct bytes (`error`, BASELINE configs[0]): the persistent kernel's literal mode, or the chain kernel
    static const bool literalOk = getenv("CXG_NO_LITERAL_KERNEL") == nullptr;
    litKernel = literalOk && !fieldsKernel && !submatch && !denseChain && !(h->flags & (cxgdev::kFlagChainBounded | cxgdev::kFlagChainSets)) &&
                cxgdev::literal_shape(*reinterpret_cast<const cxgdev::ChainAux*>(a.chain)) != 0;
    // run a run b run programs (`(\\w+)@(\\w+)\\.(\\w+)`, BASELINE configs[4]): spans, or the capture slots when every slot is the
    // start, the end or the end of the first / second run plus a constant (ChainCaps)
    static const bool trioOk = getenv("CXG_NO_TRIO_KERNEL") == nullptr;
    const int trioShape = cxgdev::trio_shape(*reinterpret_cast<const cxgdev::ChainAux*>(a.chain));
    if (!fieldsKernel && !litKernel && trioOk && !denseChain && !(h->flags & cxgdev::kFlagChainBounded) && trioShape != 0) {
      bool ok = !submatch || a.out == nullptr || fusedCaps;
      // spans with one separator for every link are the fields kernel's where it can serve the chain (with CXG_NO_FIELDS_KERNEL
      // the chain kernel's: the A/B of tests/test_gpu_fields.py); a set class (`\w+@\w+@\w+`) stays here, on the EQ instantiation
      if ((trioShape & 8) && !submatch && fieldsCould) ok = false;
      if (fusedCaps) {
        const cxgdev::ChainCaps* cc = reinterpret_cast<const cxgdev::ChainCaps*>(a.caps);
        for (uint32_t i = 0; i < cc->nruns && i < static_cast<uint32_t>(cxgdev::kCapMaxRuns); i++) ok = ok && (cc->run_op[i] & 1u) == 0u && cc->run_op[i] <= 6u;
        for (uint32_t k = 0; k < cc->nslots; k++) ok = ok && (cc->src[k] <= cxgdev::kCapSrcEnd || (cc->src[k] >= cxgdev::kCapSrcRun0 && cc->src[k] < cxgdev::kCapSrcRun0 + cc->nruns));
        ok = ok && (a.row_width & 1u) == 0u && a.row_width <= 128u && cc->nslots == a.row_width;   // <= 64 lanes write a row
      }
      trioKernel = ok;
    }
    // ... on a persistent grid with the ordering of the rows deferred by a round (k_scan_fields_pers) unless FindAll has an n
    // (the early stop lives in the grouped kernel's look-back), the phase profile is on, or a watchdog ever fired
    static const bool persOk = getenv("CXG_NO_PERSIST") == nullptr;
    a.pf_status = nullptr; a.pf_cap = 0; a.pf_epoch = 0; a.pf_full = a.pf_tpw_last = a.pf_units_last = 0;
    a.pf_rec = nullptr; a.pf_rec_rounds = 0; a.pf_stats = nullptr; a.pf_ticket = nullptr; a.pf_ncounters = 0;
    // (TRIO mode: built and measured in round 5 — config 5 0.438 ms against 0.395 on the grouped kernel, `(\d+)\.(\d+)\.(\d+)\.(\d+)` 0.57
    // against 0.44: that mathematics is VALU- and LDS-bound and the persistent instantiation holds half the waves — so off unless asked for)
    static const bool trioPers = getenv("CXG_TRIO_PERS") != nullptr;
    bool persWanted = (fieldsKernel || litKernel || (trioKernel && trioPers)) && persOk && a.static_groups && a.limit == 0 && a.prof == nullptr && a.dbg == 0 && !persDenied;
    if (persWanted && !ps.persistent.allowed()) { ps.persistent.consume(); persDenied = true; persWanted = false; }
    if (persWanted) {
      const uint64_t nwt = (len + cxgdev::kWaveTile - 1) / cxgdev::kWaveTile;
      const uint64_t need = nwt / 4u + 2u * 8192u + 64u;                                       // (full rounds + 1) x W unit words, W <= 8192 waves
      const uint64_t rneed = nwt / (4u * 1024u) + 8u;                                          // rounds: >= 1024 waves on a long haystack (+ the tail's small units)
      bool fresh = false;
      if (need > s.pfCap) {
        if (s.pfStatus) HIP_TRY(hipFree(s.pfStatus));
        s.pfStatus = nullptr; s.pfCap = 0;
        const uint64_t c = need + need / 4;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.pfStatus), c * sizeof(uint32_t)));
        s.pfCap = c; fresh = true;
      }
      if (rneed > s.pfRecRounds) {
        if (s.pfRec) HIP_TRY(hipFree(s.pfRec));
        s.pfRec = nullptr; s.pfRecRounds = 0;
        const uint64_t c = rneed + rneed / 4;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.pfRec), c * cxgdev::kPfRecStride * 8u));
        s.pfRecRounds = c; fresh = true;
      }
      if (!s.pfTickets) { HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.pfTickets), 32 * 64 * cxgdev::kPfCtrStride * sizeof(uint32_t))); fresh = true; }
      if (fresh || s.pfEpoch >= 0xFFFFu) {                                                     // all three arrays carry the same epoch
        HIP_TRY(hipMemsetAsync(s.pfTickets, 0, 32 * 64 * cxgdev::kPfCtrStride * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(s.pfStatus, 0, s.pfCap * sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(s.pfRec, 0, s.pfRecRounds * cxgdev::kPfRecStride * 8u, stream));
        s.pfEpoch = 0;
      }
      if (!s.pfStats) { HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.pfStats), 4 * 8192 * sizeof(uint64_t))); HIP_TRY(hipMemsetAsync(s.pfStats, 0, 4 * 8192 * sizeof(uint64_t), stream)); }
      a.pf_status = s.pfStatus; a.pf_cap = s.pfCap; a.pf_epoch = ++s.pfEpoch;
      a.pf_rec = s.pfRec; a.pf_rec_rounds = s.pfRecRounds; a.pf_stats = s.pfStats; a.pf_ticket = s.pfTickets;
    }
    if (a.pf_status == nullptr) litKernel = false;                  // no persistent launch for this call: the chain kernel
    static const bool countSumOk = getenv("CXG_NO_COUNT_SUM") == nullptr;
    a.count_sum = ((fieldsKernel || litKernel || (trioKernel && a.pf_status != nullptr)) && countSumOk && a.out == nullptr && a.max_len == 0 && a.limit == 0 && a.prof == nullptr && !a.dbg) ? 1u : 0u;
    if (a.u32_rows && a.out != nullptr && !((fieldsKernel || litKernel) && a.pf_status != nullptr)) {
      // the persistent kernel has the compact epilogue; when THIS call cannot have it (a rerun, the mode demoted for a while, FindAll's n) the
      // caller uses cxg_find_all_device for the call (CXG_E_INPUT), the program itself stays served
      const bool couldPers = (fieldsCould && fieldsOk && !submatch) || cxgdev::literal_shape(*reinterpret_cast<const cxgdev::ChainAux*>(a.chain)) != 0;
      return fail((relaunches || couldPers) ? CXG_E_INPUT : CXG_E_UNSUPPORTED, couldPers ? "compact rows (cxg_find_all_device_u32): the persistent kernel is not available for this call (demoted after a watchdog hit, FindAll with an n, or match-dense input): use cxg_find_all_device"
                                                                                            : "compact rows (cxg_find_all_device_u32): this program's span kernel writes int64 rows only");
    }
    if (trioKernel) {                                               // the field class as a class plan (wave_common.hpp)
      const cxgdev::ChainAux* tc = reinterpret_cast<const cxgdev::ChainAux*>(a.chain);
      uint8_t lo1[4] = {0, 0, 0, 0}, hi1[4] = {0, 0, 0, 0};
      uint32_t n1 = 1;
      if (tc->cls_kind[0] == cxgdev::kClsSet) { n1 = tc->cls_nr[0]; for (uint32_t q = 0; q < 4; q++) { lo1[q] = tc->cls_rlo[0][q]; hi1[q] = tc->cls_rhi[0][q]; } }
      else if (tc->cls_kind[0] == cxgdev::kClsDigit) { lo1[0] = 0x30; hi1[0] = 0x39; }
      else { lo1[0] = tc->cls_lo[0]; hi1[0] = tc->cls_hi[0]; }
      static const bool plansOk = getenv("CXG_NO_PLANS") == nullptr;
      a.plan = cxgdev::plan_class(n1, lo1, hi1);
      a.plan_shape = plansOk ? static_cast<uint32_t>(cxgdev::plan_shape(a.plan)) : 0u;
    }
    le = hipSuccess;
    if (litKernel) {                                                // (a launch the persistent geometry cannot hold: the chain kernel below)
      le = cxgdev::launch_scan_fields_wave(a, stream, &persKernel);
      if (!persKernel) { litKernel = false; a.count_sum = 0; (void)hipGetLastError(); }
    }
    if (litKernel) {}
    else if (trioKernel) le = cxgdev::launch_scan_trio_wave(a, stream, &persKernel);   // (the grouped kernel ignores count_sum: its look-back leaves the total)
    else if (fieldsKernel) le = cxgdev::launch_scan_fields_wave(a, stream, &persKernel);
    else le = cxgdev::launch_scan_chain_wave(a, reinterpret_cast<const cxgdev::ChainAux*>(hb + h->aux_off + 256)->ncls,
                                        (h->flags & cxgdev::kFlagChainSets) != 0, fusedCaps, stream);
  }
  else switch (h->kind) {
    case cxgdev::kKindDigit:
      if (gen == 1) le = cxgdev::launch_scan_dfa(h->kind, a, h->fwd_states, h->rev_states, stream);
      else le = cxgdev::launch_scan_digit_flat(a, h->fwd_states, stream);
      break;
    case cxgdev::kKindBidir: le = cxgdev::launch_scan_dfa(h->kind, a, h->fwd_states, h->rev_states, stream); break;
    case cxgdev::kKindCharClass: le = cxgdev::launch_scan_charclass(a, stream); break;
    case cxgdev::kKindTeddy: le = cxgdev::launch_scan_teddy(a, stream); break;
    default: return fail(CXG_E_INTERNAL, "unknown program kind");
  }
  if (le != hipSuccess) return failHip(le, "kernel launch");
  const uint32_t kernelId = static_cast<uint32_t>(gen == 11 ? CXG_K_DELIM_WAVE : trioKernel ? (persKernel ? CXG_K_TRIO_PERS : CXG_K_TRIO_WAVE) : litKernel ? CXG_K_LITERAL_PERS : persKernel ? CXG_K_FIELDS_PERS : fieldsKernel ? CXG_K_FIELDS_WAVE : (gen == 10 && fsmDirectRan) ? (fsmDirectTables ? CXG_K_FSM_DIRECT : CXG_K_FSM_LEAN) : gen >= 6 ? gen
                                                  : h->kind == cxgdev::kKindDigit ? (gen == 1 ? CXG_K_DFA_TABLE : CXG_K_DIGIT_FLAT)
                                                  : h->kind == cxgdev::kKindBidir ? CXG_K_DFA_TABLE : h->kind == cxgdev::kKindTeddy ? CXG_K_TEDDY_TABLE : CXG_K_CHARCLASS_TABLE);
  if (nladder < sizeof ladder) ladder[nladder] = static_cast<uint8_t>(kernelId);
  nladder++;
  uint32_t launches = 1;
  if (goAsync) {                                                    // cxg_find_all_device_async: the launch stays in flight, cxg_wait finishes the call
    HIP_TRY(hipEventRecord(as->ev[1], stream));
    as->stream = stream; as->kernelId = kernelId; as->tiles = a.ntiles;
    as->mode = gen == 11 ? 3u : persKernel ? 2u : a.static_groups ? 1u : 0u;
    orderGate.close();
    s.asyncInFlight++;
    return kRcPending;
  }
  if (submatch && a.out && !fusedCaps) { if (int rc = launchCapturePass(p, s, a, d_cap, stream, launches)) return rc; }
  if (wantEv) HIP_TRY(hipEventRecord(s.ev[2], stream));
  if (!a.epoch) HIP_TRY(hipMemcpyAsync(s.hostCtl, s.ctl, 32, hipMemcpyDeviceToHost, stream));   // wave kernels wrote hostCtl themselves
  orderGate.close();
  HIP_TRY(syncStream(stream));
  const uint64_t total = s.hostCtl[1];
  uint32_t err = static_cast<uint32_t>(s.hostCtl[2]);
  if (timing) {
    float k = 0, t = 0;
    (void)hipEventElapsedTime(&k, s.ev[1], s.ev[2]);
    if (ev0) (void)hipEventElapsedTime(&t, s.ev[0], s.ev[2]); else t = k;
    timing->kernel_ms = k; timing->total_ms = t; timing->n_launches = launches + relaunches;
    timing->n_ladder = nladder;
    std::memcpy(timing->ladder, ladder, sizeof ladder);
    timing->grid = static_cast<uint32_t>(a.ntiles); timing->block = cxgdev::kThreads; timing->tiles = a.ntiles;
    timing->kernel = kernelId;
    timing->fallback_reason = lastReason;
  }
  if (profOn) {
    uint64_t pc[16];
    HIP_TRY(hipMemcpy(pc, s.prof, 128, hipMemcpyDeviceToHost));
    if (fieldsKernel && pc[4]) fprintf(stderr, "[CXG_PROF] fields kernel, wave 0, cycles per workgroup (%llu workgroups): tile loop %llu, first barrier %llu, prefix + look-back %llu\n",
                                       (unsigned long long)pc[4], (unsigned long long)(pc[1] / pc[4]), (unsigned long long)(pc[2] / pc[4]), (unsigned long long)(pc[3] / pc[4]));
    if (gen == 6 && pc[15]) {
      fprintf(stderr, "[CXG_PROF] gen6 waves=%llu avg cycles per wave and group:", (unsigned long long)pc[15]);
      static const char* names[7] = {"A", "ldsT", "own", "B", "starts", "F", "rows"};
      for (int i = 0; i < 7; i++) fprintf(stderr, " %s=%llu", names[i], (unsigned long long)(pc[8 + i] / pc[15]));
      fprintf(stderr, "\n");
    }
    if (gen == 10 && pc[15]) {
      fprintf(stderr, "[CXG_PROF] fsm waves=%llu avg cycles per wave and group:", (unsigned long long)pc[15]);
      static const char* names[7] = {"stage", "entry", "walk", "finish", "gather", "starts", "-"};
      for (int i = 0; i < 6; i++) fprintf(stderr, " %s=%llu", names[i], (unsigned long long)(pc[8 + i] / pc[15]));
      fprintf(stderr, "\n");
    }
    if (kernelId == CXG_K_CHARCLASS_WAVE) {                          // one record of timestamps (shader clock) per workgroup, wave 0
      const size_t ng = a.ngroups < Scratch::kProfRecords ? static_cast<size_t>(a.ngroups) : Scratch::kProfRecords;
      std::vector<uint64_t> rec(ng * 8);
      HIP_TRY(hipMemcpy(rec.data(), s.prof + 16, ng * 64, hipMemcpyDeviceToHost));
      uint64_t t0 = ~0ull, t1 = 0;
      double ph[5] = {0, 0, 0, 0, 0};
      size_t n = 0;
      for (size_t g = 0; g < ng; g++) {
        const uint64_t* r = &rec[g * 8];
        if (!r[0]) continue;
        n++;
        if (r[0] < t0) t0 = r[0];
        if (r[5] > t1) t1 = r[5];
        for (int i = 0; i < 5; i++) ph[i] += static_cast<double>(r[i + 1] - r[i]);
      }
      if (n) {
        fprintf(stderr, "[CXG_PROF] charclass: %zu workgroups, wave 0, shader-clock cycles per workgroup: claim+issue %.0f, pass 1 %.0f, barrier %.0f, prefix+look-back %.0f, pass 2 %.0f; "
                        "first start to last end %llu cycles; starts of workgroups 0 / 1023 / 1024 / 2048 / 8192 after the first: %llu %llu %llu %llu %llu\n",
                n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, (unsigned long long)(t1 - t0),
                (unsigned long long)(rec[0] - t0), (unsigned long long)(ng > 1023 ? rec[1023 * 8] - t0 : 0), (unsigned long long)(ng > 1024 ? rec[1024 * 8] - t0 : 0),
                (unsigned long long)(ng > 2048 ? rec[2048 * 8] - t0 : 0), (unsigned long long)(ng > 8192 ? rec[8192 * 8] - t0 : 0));
      }
    }
    if (gen == 6 && pc[7])
      fprintf(stderr, "[CXG_PROF] gen6 pairing mismatch: tile_lo=%llu n=%llu n_ends=%llu cout=%llu zA=%lld zB=%lld stage=%llu (count %llu)\n",
              (unsigned long long)pc[0], (unsigned long long)pc[1], (unsigned long long)pc[2], (unsigned long long)pc[3],
              (long long)pc[4], (long long)pc[5], (unsigned long long)pc[6], (unsigned long long)pc[7]);
    else if (pc[5])
      fprintf(stderr, "[CXG_PROF] waves=%llu avg cycles/wave: tables=%llu tile=%llu walk=%llu scan=%llu lookback=%llu\n",
              (unsigned long long)pc[5], (unsigned long long)(pc[0] / pc[5]), (unsigned long long)(pc[1] / pc[5]),
              (unsigned long long)(pc[2] / pc[5]), (unsigned long long)(pc[3] / pc[5]), (unsigned long long)(pc[4] / pc[5]));
  }
  if (a.pf_status) {
    static const bool pfVerbose = getenv("CXG_VERBOSE") != nullptr;
    if (pfVerbose) {                                                // units that had to wait for their round's record, polls
      std::vector<uint64_t> st(4 * 8192);
      HIP_TRY(hipMemcpy(st.data(), s.pfStats, st.size() * 8, hipMemcpyDeviceToHost));
      HIP_TRY(hipMemset(s.pfStats, 0, st.size() * 8));
      uint64_t w = 0, pl = 0, mx = 0, nw = 0;
      double lifeX[8] = {0}, scanX[8] = {0}, lifeMaxX[8] = {0}; uint64_t nX[8] = {0};
      std::vector<uint64_t> lives, scans;
      for (size_t i = 0; i < 8192; i++) {
        w += st[i] >> 32; pl += st[i] & 0xFFFFFFFFull; mx = std::max<uint64_t>(mx, st[i] & 0xFFFFFFFFull);
        if (!st[8192 + i]) continue;
        nw++; lives.push_back(st[8192 + i]); scans.push_back(st[16384 + i]);
        const int x = static_cast<int>(st[24576 + i] >> 32) & 7;
        lifeX[x] += st[8192 + i]; scanX[x] += st[16384 + i]; nX[x]++; lifeMaxX[x] = std::max<double>(lifeMaxX[x], st[8192 + i]);
      }
      fprintf(stderr, "[cxg] persistent fields kernel: %llu units waited for their round's record, %llu polls (most by one wave: %llu)\n", (unsigned long long)w, (unsigned long long)pl, (unsigned long long)mx);
      if (nw) {
        std::sort(lives.begin(), lives.end()); std::sort(scans.begin(), scans.end());
        auto q = [&](const std::vector<uint64_t>& v, double f) { return v[static_cast<size_t>(f * (v.size() - 1))] / 1000.0; };
        fprintf(stderr, "[cxg]   %llu waves; life in 1000 s_memtime ticks (~2.2 GHz in a busy kernel) min/p10/median/p90/max %.1f %.1f %.1f %.1f %.1f; in tile loops %.1f %.1f %.1f %.1f %.1f\n", (unsigned long long)nw,
                q(lives, 0), q(lives, 0.1), q(lives, 0.5), q(lives, 0.9), q(lives, 1), q(scans, 0), q(scans, 0.1), q(scans, 0.5), q(scans, 0.9), q(scans, 1));
        for (int x = 0; x < 8; x++) if (nX[x]) fprintf(stderr, "[cxg]   XCD %d: %llu waves, life mean %.1f max %.1f k ticks, tile loops mean %.1f k ticks\n", x, (unsigned long long)nX[x], lifeX[x] / nX[x] / 1000.0, lifeMaxX[x] / 1000.0, scanX[x] / nX[x] / 1000.0);
      }
    }
  }
  if (err & 2u) {
    static const bool wdVerbose = getenv("CXG_VERBOSE") != nullptr;
    const uint32_t origin = (err >> 24) & 15u;
    if (wdVerbose) fprintf(stderr, "[cxg] spin watchdog fired (wait %u, kernel %u, static groups %u): this call reruns one mode down\n", origin, kernelId, a.static_groups);
    if (gen == 11) {                                                // the delimiter kernel needs dispatch in index order: the transducer for a while
      ps.delim.demote();
      relaunches++; gen = 10; fsmTried = true; continue;
    }
    if (persKernel) {                                               // the persistent grid was not co-resident: the grouped kernel, still with static groups
      ps.persistent.demote();
      persDenied = true; relaunches++; continue;
    }
    if (a.static_groups) {                                          // dispatch was not in index order: tickets
      ps.staticGroups.demote();
      staticDenied = true; relaunches++; continue;
    }
  } else {
    if (gen == 11) ps.delim.clean();
    if (persKernel) ps.persistent.clean();
    else if (a.static_groups) ps.staticGroups.clean();
  }
  err &= 0x00FFFFFFu;
  if ((err & 8u) && gen >= 3) {
    static const bool verbose = getenv("CXG_VERBOSE") != nullptr;
    if (gen == 10 && fsmDirectRan && ((err >> 8) & ~0x72u) != 0u) {   // the lean kernel: an entry state that did not collapse, a match pending past the window — k_scan_fsm has the machinery
      if (verbose) fprintf(stderr, "[cxg] transducer kernel, lean form: reason bits 0x%x, rerunning on k_scan_fsm\n", err >> 8);
      fsmDirect = false;
      if ((err >> 8) & 1u) p->fsmNoDirect[submatch ? 1 : 0].store(1, std::memory_order_relaxed);   // (input without synchronising structure: remembered for the program)
      relaunches++;
      continue;
    }
    if (gen == 10 && ((err >> 8) & 0x32u) != 0u && ((err >> 8) & ~0x72u) == 0u && fsmMode < 2) {   // transducer kernel: row / event buffers overflowed
      // (0x40 — a row without a start — beside an overflow bit is a consequence of the dropped rows, not a finding)
      // 0x20 alone: the wave's row list -> mode 1 (2 tiles per wave); a sub-chunk's own buffers (0x02 rows, 0x10 events), or
      // mode 1 was not enough -> mode 2 (1 tile, 2048 rows, 16 rows / 32 events per 32 bytes)
      fsmMode = ((err >> 8) == 0x20u && fsmMode == 0) ? 1 : 2;
      if (verbose) fprintf(stderr, "[cxg] transducer kernel: match-dense input (reason bits 0x%x), rerunning in mode %d\n", err >> 8, fsmMode);
      {                                                             // remembered per program; only grows
        uint8_t old = p->fsmMode[submatch ? 1 : 0].load(std::memory_order_relaxed);
        while (old < fsmMode && !p->fsmMode[submatch ? 1 : 0].compare_exchange_weak(old, static_cast<uint8_t>(fsmMode), std::memory_order_relaxed)) {}
      }
      relaunches++;
      continue;
    }
    if ((gen == 6 || gen == 7 || gen == 9) && (err >> 8) == 0x10u && !denseChain && !(h->flags & cxgdev::kFlagChainBounded)) {   // only the row buffers overflowed: same kernel, two tiles per wave
      if (verbose) fprintf(stderr, "[cxg] wave kernel: row buffers overflowed (match-dense input), rerunning with %d tiles per wave\n", cxgdev::kDenseTilesPerWave);
      denseChain = true;
      p->denseChain[submatch ? 1 : 0].store(1, std::memory_order_relaxed);
      if (fsmMode == 0) fsmMode = 1;                                // (should this call still reach the transducer: the input is match-dense)
      relaunches++;
      continue;
    }
    lastReason = err >> 8;
    if (d_fsm && !fsmTried) {                                       // dense tile / no sync byte in a halo: the transducer kernel
      if (verbose) fprintf(stderr, "[cxg] gen %d raised the fallback flag (reason bits 0x%x): rerunning with the transducer kernel\n", gen, err >> 8);
      relaunches++; gen = 10; fsmTried = true; continue;
    }
    if (h->kind == cxgdev::kKindFsmOnly)                            // no table-walking image: degrade for THIS haystack
      return fail(CXG_E_INPUT, "haystack outside the transducer kernel's budgets (reason bits " + std::to_string(err >> 8) +
                               "): matches denser than one per 2 bytes, a match reaching > 190 bytes past its tile, or an unresolvable entry state");
    if (h->kind == cxgdev::kKindTeddy && !submatch && reinterpret_cast<const cxgdev::TeddyAux*>(p->blob.data() + h->aux_off)->looks != 0u)
      return fail(CXG_E_INPUT, "haystack outside the literal kernel's and the transducer kernel's budgets (reason bits " + std::to_string(err >> 8) + "); the table kernel knows no assertions");
    if (h->kind == cxgdev::kKindCharClass && (h->flags & cxgdev::kFlagCcRanges) && reinterpret_cast<const cxgdev::CharClassAux*>(p->blob.data() + h->aux_off)->pairs)
      return fail(CXG_E_INPUT, "more than 1024 occurrences of the quote byte in one 3840-byte tile (no table kernel pairs them)");
    if (verbose) fprintf(stderr, "[cxg] gen %d raised the fallback flag (reason bits 0x%x): rerunning with the table kernel\n", gen, err >> 8);
    relaunches++; gen = h->kind == cxgdev::kKindDigit ? 2 : 0; continue;   // table-walking kernels: exact, serial inside a stretch
  }
  err &= 0xFFu;
  // (first: a walk cut at the budget leaves a truncated row behind, which may also have raised the long-match flag — the rows
  // of such a launch are not the reference's and must not reach the UseBoth restart loop)
  if (err & cxgdev::kErrSerialLimit)
    return fail(CXG_E_INPUT, "haystack has a stretch without synchronising bytes beyond the serial-walk budget (128 KiB)");
  if (err & cxgdev::kErrLongMatch) {
    if (n_out) *n_out = total;
    return kRcLongMatch;
  }
  if (err) return fail(CXG_E_INTERNAL, "device-side watchdog/overflow flag " + std::to_string(err));
  if (dbgBits) { if (n_out) *n_out = total; return CXG_OK; }
  uint64_t n = total;
  if (limit > 0 && n > static_cast<uint64_t>(limit)) n = static_cast<uint64_t>(limit);
  if (n_out) *n_out = n;
  if (d_out && n > cap) return fail(CXG_E_CAPACITY, "output capacity too small");
  (void)row_width;
  return CXG_OK;
  }   // one span launch
}

__global__ void k_first_long(const int64_t* rows, uint64_t n, uint32_t width, int64_t max_len, unsigned long long* first) {
  for (uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    if (rows[i * width + 1] - rows[i * width] > max_len) atomicMin(first, static_cast<unsigned long long>(i));
}

// UseBoth (findIndicesAdaptiveAtWithState, meta/find_indices.go:408-441): the DFA's match end `end` only picks where the
// PikeVM starts — at the search position `at`, or at end - 100 when end > at + 100.  Nothing matches between `at` and the
// leftmost match, so the PikeVM's answer is the plain leftmost-first match unless that match is longer than 100 bytes; then
// the PikeVM starts INSIDE it and FindAll continues with whatever it finds from there.  On the device: the kernels iterate
// plain leftmost-first and flag a longer match; every row in front of the first such match stands, and the search restarts
// where the reference's PikeVM would — at that match's end - 100 — on an aligned copy of the haystack's suffix, with `base`
// moved accordingly.  Each restart begins behind the start of the match that caused it, so the loop ends; more than
// kMaxBothRestarts long matches in one haystack are refused (CXG_E_INPUT, the caller keeps its CPU loop).
constexpr int kMaxBothRestarts = 64;
int scanDevice(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out,
               uint64_t cap, uint64_t* n_out, void* user_stream, cxg_timing* timing, int row_width) {
  if (p && p->nullable && row_width == 2 && p->supported) return scanNullable(p, d_hay, len, base, limit, d_out, cap, n_out, user_stream, timing);
  if (p && p->nullable && row_width > 2 && p->subNullable && p->supported) return scanNullableSubmatch(p, d_hay, len, base, limit, d_out, cap, n_out, user_stream, timing, row_width);
  // capture slots at fixed distances from the span's ends: FindAll + one expansion kernel (no capture pass per row), unless the
  // chain kernels write the slots themselves
  static const bool offCapsOk = getenv("CXG_NO_OFFSET_CAPS") == nullptr;
  if (p && row_width > 2 && p->offCapsOn && p->supported && offCapsOk && !(p->subSupported && p->chainCaps[0]))
    return scanOffsetCaps(p, d_hay, len, base, limit, d_out, cap, n_out, user_stream, timing, row_width);
  uint64_t n_cur = 0;
  int rc = scanDeviceOnce(p, d_hay, len, base, limit, d_out, cap, &n_cur, user_stream, timing, row_width);
  if (rc != kRcLongMatch) { if (n_out) *n_out = n_cur; return rc; }
  const bool submatch = row_width > 2;
  const cxgdev::BlobHeader* h = reinterpret_cast<const cxgdev::BlobHeader*>(submatch ? p->subBlob.data() : p->blob.data());
  const auto* fh = reinterpret_cast<const cxgdev::FsmHeader*>((submatch ? p->subFsmBlob : p->fsmBlob).data());
  const bool look = !(submatch ? p->subFsmBlob : p->fsmBlob).empty() && fh->nk > 1;
  (void)h;
  if (look)   // the restarted search would need the byte in front of its first one as context
    return fail(CXG_E_INPUT, "UseBoth program with assertions met a match longer than 100 bytes (the reference restarts its PikeVM inside such a match)");
  Scratch* sp;
  if (int r = getScratch(&sp)) return r;
  Scratch& s = *sp;
  hipStream_t stream = user_stream ? static_cast<hipStream_t>(user_stream) : s.stream;
  if (!s.bothFirst) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.bothFirst), 16));
  cxg_timing acc;
  std::memset(&acc, 0, sizeof acc);
  auto add_timing = [&]() {
    if (!timing) return;
    acc.kernel_ms += timing->kernel_ms; acc.total_ms += timing->total_ms; acc.n_launches += timing->n_launches;
    acc.grid = timing->grid; acc.block = timing->block; acc.tiles = timing->tiles; acc.kernel = timing->kernel; acc.fallback_reason = timing->fallback_reason;
    for (uint32_t i = 0; i < timing->n_ladder && i < sizeof timing->ladder; i++) { if (acc.n_ladder < sizeof acc.ladder) acc.ladder[acc.n_ladder] = timing->ladder[i]; acc.n_ladder++; }
  };
  add_timing();
  const uint64_t width = static_cast<uint64_t>(row_width);
  int64_t* const out = static_cast<int64_t*>(d_out);
  uint64_t done = 0;                       // rows that stand
  uint64_t abs_off = 0;                    // where the current search started, in the caller's haystack
  const uint8_t* cur = static_cast<const uint8_t*>(d_hay);
  for (int iter = 0; iter < kMaxBothRestarts; iter++) {
    // the rows of the launch that met the long match
    const uint64_t room = out ? (cap > done ? cap - done : 0) : 0;
    const int64_t lim_rem = limit > 0 ? limit - static_cast<int64_t>(done) : limit;
    const int64_t* rows = out ? out + done * width : nullptr;
    uint64_t nscan = n_cur;                                        // rows that matter: FindAll(n) stops after n of them
    if (lim_rem > 0 && nscan > static_cast<uint64_t>(lim_rem)) nscan = static_cast<uint64_t>(lim_rem);
    if (room < nscan) {
      if (nscan * width > s.bothRowsCap) {
        if (s.bothRows) HIP_TRY(hipFree(s.bothRows));
        s.bothRows = nullptr; s.bothRowsCap = 0;
        const uint64_t c = nscan * width + nscan * width / 4 + 1024;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.bothRows), c * sizeof(int64_t)));
        s.bothRowsCap = c;
      }
      uint64_t n2 = 0;
      rc = scanDeviceOnce(p, cur, len - abs_off, base + static_cast<int64_t>(abs_off), -1, s.bothRows, nscan, &n2, user_stream, timing, row_width);
      add_timing();
      // (a launch WITH a limit lets groups behind the n-th row publish `limit` instead of their own count — block_common.hpp
      // limit_reached_skip — so its total is a lower bound of the unlimited rerun's; only the first nscan rows are used)
      const bool agrees = lim_rem > 0 ? n2 >= nscan : n2 == n_cur;
      if (rc != kRcLongMatch || !agrees) return rc == kRcLongMatch || rc == CXG_OK ? fail(CXG_E_INTERNAL, "UseBoth restart: the rerun for rows disagrees with the count") : rc;
      rows = s.bothRows;
    }
    OrderGate restartGate(g_path[s.device < 0 ? 0 : s.device], stream);   // (this helper kernel ran outside the device's launch order)
    HIP_TRY(hipMemsetAsync(s.bothFirst, 0xFF, 8, stream));
    const uint32_t blocks = static_cast<uint32_t>(std::min<uint64_t>((nscan + 255) / 256, 4096));
    // The first row of a RESTARTED search is what the reference's PikeVM returned from end - 100: it stands whatever its length
    // (the next match downstream can be a long one, reported in full); the 100-byte rule applies to the searches behind it.
    const uint64_t skip = (iter > 0 && nscan > 0) ? 1 : 0;
    hipLaunchKernelGGL(k_first_long, dim3(blocks), dim3(256), 0, stream, rows + skip * width, nscan - skip, static_cast<uint32_t>(row_width), static_cast<int64_t>(cxgdev::kBothRestartSpan), s.bothFirst);
    unsigned long long k = 0;
    HIP_TRY(hipMemcpyAsync(&k, s.bothFirst, 8, hipMemcpyDeviceToHost, stream));
    restartGate.close();
    HIP_TRY(hipStreamSynchronize(stream));
    k = k >= nscan - skip ? nscan : k + skip;
    bool over_estimate = false;
    if (k >= nscan) {
      // no long row among them: the long match lies behind the n-th row (the first n stand), or the kernel's flag was an
      // over-estimate — the transducer kernel measures the first row of a group before its start is bounded by the previous
      // row (k_fsm_fix_heads corrects the row afterwards): every row of the launch stands
      over_estimate = nscan == n_cur;
      k = nscan;
    }
    int64_t e = 0;
    if (k < nscan) {
      HIP_TRY(hipMemcpyAsync(&e, rows + k * width + 1, 8, hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
    }
    if (rows == s.bothRows && out && room) {                       // the rows that stand, as far as the caller has room
      const uint64_t ncopy = std::min<uint64_t>(k, room);
      if (ncopy) HIP_TRY(hipMemcpyAsync(out + done * width, s.bothRows, ncopy * width * sizeof(int64_t), hipMemcpyDefault, stream));
    }
    done += k;
    if (over_estimate) { n_cur = 0; rc = CXG_OK; break; }
    if (limit > 0 && done >= static_cast<uint64_t>(limit)) { n_cur = 0; done = static_cast<uint64_t>(limit); rc = CXG_OK; break; }
    (void)lim_rem;
    // restart where the reference's PikeVM starts: end - 100 (absolute), on an aligned copy of the suffix
    const uint64_t e_abs = static_cast<uint64_t>(e - base);
    const uint64_t next = e_abs - cxgdev::kBothRestartSpan;
    if (next <= abs_off) return fail(CXG_E_INTERNAL, "UseBoth restart does not advance");
    const uint64_t rest = len - next;
    if (rest + 64 > s.bothHayCap) {
      if (s.bothHay) HIP_TRY(hipFree(s.bothHay));
      s.bothHay = nullptr; s.bothHayCap = 0;
      const uint64_t c = rest + 4096;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.bothHay), c));
      s.bothHayCap = c;
    }
    HIP_TRY(hipMemcpyAsync(s.bothHay, static_cast<const uint8_t*>(d_hay) + next, rest, hipMemcpyDefault, stream));
    HIP_TRY(hipMemsetAsync(s.bothHay + rest, 0, 64, stream));
    abs_off = next;
    cur = s.bothHay;
    const uint64_t room2 = out ? (cap > done ? cap - done : 0) : 0;
    rc = scanDeviceOnce(p, cur, rest, base + static_cast<int64_t>(abs_off), limit > 0 ? limit - static_cast<int64_t>(done) : limit,
                        room2 ? out + done * width : nullptr, room2, &n_cur, user_stream, timing, row_width);
    add_timing();
    if (rc == kRcLongMatch) continue;
    if (rc == CXG_E_CAPACITY) { done += n_cur; n_cur = 0; }
    break;
  }
  if (timing) *timing = acc;
  // the restart loop's own staging follows the rule of s.hay / s.out: at most kKeepStagingBytes stay with the thread
  if (s.bothHayCap > kKeepStagingBytes || s.bothRowsCap * sizeof(int64_t) > kKeepStagingBytes) {
    (void)hipStreamSynchronize(stream);
    if (s.bothHayCap > kKeepStagingBytes) { (void)hipFree(s.bothHay); s.bothHay = nullptr; s.bothHayCap = 0; }
    if (s.bothRowsCap * sizeof(int64_t) > kKeepStagingBytes) { (void)hipFree(s.bothRows); s.bothRows = nullptr; s.bothRowsCap = 0; }
  }
  if (rc == kRcLongMatch) return fail(CXG_E_INPUT, "UseBoth program met more than 64 matches longer than 100 bytes in one haystack");
  if (rc != CXG_OK && rc != CXG_E_CAPACITY) return rc;
  const uint64_t n = done + n_cur;
  if (n_out) *n_out = n;
  if (out && n > cap) return fail(CXG_E_CAPACITY, "output capacity too small");
  return CXG_OK;
}

// ---- nullable programs -------------------------------------------------------------------------------------------------------
// FindAll of a pattern that matches the empty string (meta/findall.go:216-283): the rows R of its non-empty variant
// (program.cc nonEmptyVariant), and an empty match [p, p] at every position p in 0..len outside the closed intervals [s, e] of
// R — inside a match the loop does not search, at its end the empty match is skipped (`start == end && start == lastMatchEnd`,
// :251-257), everywhere else the search at p answers at once with the empty path.  All in position order.
// cov[i] = size of the union of the closed intervals of rows 0..i (adjacent rows share their common point).
constexpr uint32_t kNullBlock = 4096;                              // rows per block of the prefix sum
__global__ __launch_bounds__(1024) void k_null_cover(const int64_t* rows, uint64_t n, uint64_t* cov, uint64_t* bsum) {
  __shared__ uint64_t s_w[16];
  const uint64_t b0 = static_cast<uint64_t>(blockIdx.x) * kNullBlock;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint64_t c[4], t = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t i = b0 + static_cast<uint64_t>(threadIdx.x) * 4 + k;
    c[k] = 0;
    if (i < n) {
      const int64_t s = rows[2 * i], e = rows[2 * i + 1];
      c[k] = static_cast<uint64_t>(e - s + 1) - ((i > 0 && rows[2 * i - 1] == s) ? 1u : 0u);
    }
    t += c[k];
    c[k] = t;                                                       // inclusive inside the thread
  }
  uint64_t incl = t;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint64_t up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint64_t off = incl - t;
  for (int w = 0; w < wave; w++) off += s_w[w];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint64_t i = b0 + static_cast<uint64_t>(threadIdx.x) * 4 + k;
    if (i < n) cov[i] = off + c[k];
  }
  if (threadIdx.x == 1023) bsum[blockIdx.x] = off + t;
}
__global__ __launch_bounds__(1024) void k_null_block_offsets(uint64_t* bsum, uint64_t nb) {   // exclusive sums of the block totals, one workgroup
  __shared__ uint64_t s_w[16];
  __shared__ uint64_t s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint64_t b0 = 0; b0 < nb; b0 += 1024) {
    const uint64_t i = b0 + threadIdx.x;
    const uint64_t v = i < nb ? bsum[i] : 0;
    uint64_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint64_t up = __shfl_up(incl, d, 64); if (lane >= d) incl += up; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    uint64_t off = s_carry + incl - v;
    for (int w = 0; w < wave; w++) off += s_w[w];
    if (i < nb) bsum[i] = off;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = off + v;
    __syncthreads();
  }
}
__device__ __forceinline__ uint64_t null_cov_incl(const uint64_t* cov, const uint64_t* bsum, uint64_t i) { return cov[i] + bsum[i / kNullBlock]; }
// the non-empty rows at their places: rows in front + uncovered positions in front
__global__ void k_null_rows(const int64_t* rows, uint64_t n, const uint64_t* cov, const uint64_t* bsum, int64_t base, int64_t* out, uint64_t cap) {
  const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t s = rows[2 * i], e = rows[2 * i + 1];
  const uint64_t adj = (i > 0 && rows[2 * i - 1] == s) ? 1u : 0u;
  const uint64_t own = static_cast<uint64_t>(e - s + 1) - adj;
  const uint64_t below = null_cov_incl(cov, bsum, i) - own - adj;   // covered positions strictly below s
  const uint64_t at = i + (static_cast<uint64_t>(s) - below);
  if (at < cap) cxgdev::store_pair_nt(out + 2 * at, base + s, base + e);
}
// the empty matches: one thread per position 0..len
__global__ void k_null_empties(const int64_t* rows, uint64_t n, const uint64_t* cov, const uint64_t* bsum, uint64_t len, int64_t base, int64_t* out, uint64_t cap) {
  const uint64_t p = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p > len) return;
  uint64_t lo = 0, hi = n;                                          // number of rows with start <= p
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (static_cast<uint64_t>(rows[2 * mid]) <= p) lo = mid + 1; else hi = mid;
  }
  uint64_t at = p;
  if (lo > 0) {
    if (p <= static_cast<uint64_t>(rows[2 * (lo - 1) + 1])) return; // inside a match, or at its end
    at = lo + (p - null_cov_incl(cov, bsum, lo - 1));
  }
  if (at < cap) cxgdev::store_pair_nt(out + 2 * at, base + static_cast<int64_t>(p), base + static_cast<int64_t>(p));
}

int scanNullable(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out, uint64_t cap,
                 uint64_t* n_out, void* user_stream, cxg_timing* timing) {
  if (n_out) *n_out = 0;
  if (timing) std::memset(timing, 0, sizeof *timing);
  if (limit == 0) return CXG_OK;
  if (len >= (1ull << 40)) return fail(CXG_E_INVALID, "haystack too large for one launch; shard it");
  Scratch* sp;
  if (int rc = getScratch(&sp)) return rc;
  Scratch& s = *sp;
  hipStream_t stream = user_stream ? static_cast<hipStream_t>(user_stream) : s.stream;
  if (d_out && (reinterpret_cast<uintptr_t>(d_out) & 15u)) return fail(CXG_E_INVALID, "device output must be 16-byte aligned");
  uint64_t n = 0;
  cxg_timing inner;
  std::memset(&inner, 0, sizeof inner);
  float kernel_ms = 0, total_ms = 0;
  uint32_t launches = 0;
  if (!p->nullableOnlyEmpty && len > 0) {
    if (int rc = scanDeviceOnce(p, d_hay, len, 0, -1, nullptr, 0, &n, user_stream, &inner, 2)) return rc;
    kernel_ms += inner.kernel_ms; total_ms += inner.total_ms; launches += inner.n_launches;
    if (n > 0) {
      if (2 * n > s.nullRowsCap) {
        if (s.nullRows) HIP_TRY(hipFree(s.nullRows));
        s.nullRows = nullptr; s.nullRowsCap = 0;
        const uint64_t c = 2 * n + n / 2 + 1024;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.nullRows), c * sizeof(int64_t)));
        s.nullRowsCap = c;
      }
      uint64_t n2 = 0;
      if (int rc = scanDeviceOnce(p, d_hay, len, 0, -1, s.nullRows, n, &n2, user_stream, &inner, 2)) return rc;
      if (n2 != n) return fail(CXG_E_INTERNAL, "nullable program: the rerun for rows disagrees with the count");
      kernel_ms += inner.kernel_ms; total_ms += inner.total_ms; launches += inner.n_launches;
    }
  }
  const uint64_t nb = (n + kNullBlock - 1) / kNullBlock;
  uint64_t covered = 0;
  OrderGate orderGate(g_path[s.device < 0 ? 0 : s.device], stream);
  HIP_TRY(hipEventRecord(s.ev[0], stream));
  if (n > 0) {
    if (n + nb + 8 > s.nullCovCap) {
      if (s.nullCov) HIP_TRY(hipFree(s.nullCov));
      s.nullCov = nullptr; s.nullCovCap = 0;
      const uint64_t c = n + nb + n / 2 + 1024;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.nullCov), c * sizeof(uint64_t)));
      s.nullCovCap = c;
    }
    uint64_t* cov = s.nullCov;
    uint64_t* bsum = s.nullCov + n;
    hipLaunchKernelGGL(k_null_cover, dim3(static_cast<unsigned>(nb)), dim3(1024), 0, stream, s.nullRows, n, cov, bsum);
    hipLaunchKernelGGL(k_null_block_offsets, dim3(1), dim3(1024), 0, stream, bsum, nb);
    HIP_TRY(hipGetLastError());
    uint64_t last[2] = {0, 0};                                      // cov[n - 1] inside its block, offset of the last block
    HIP_TRY(hipMemcpyAsync(&last[0], cov + (n - 1), 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(&last[1], bsum + (nb - 1), 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    covered = last[0] + last[1];
    launches += 2;
  }
  if (covered > len + 1) return fail(CXG_E_INTERNAL, "nullable program: rows cover more positions than the haystack has");
  uint64_t total = n + (len + 1 - covered);
  if (limit > 0 && total > static_cast<uint64_t>(limit)) total = static_cast<uint64_t>(limit);
  if (n_out) *n_out = total;
  if (d_out) {
    const uint64_t room = std::min<uint64_t>(cap, total);          // rows at places >= room are not wanted (FindAll's n) or do not fit
    int64_t* out = static_cast<int64_t*>(d_out);
    if (n > 0) hipLaunchKernelGGL(k_null_rows, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, s.nullRows, n, s.nullCov, s.nullCov + n, base, out, room);
    const uint64_t npos = len + 1;
    if ((npos + 255) / 256 > 0x7FFFFFFFull) return fail(CXG_E_INVALID, "haystack too large for one launch; shard it");
    hipLaunchKernelGGL(k_null_empties, dim3(static_cast<unsigned>((npos + 255) / 256)), dim3(256), 0, stream, s.nullRows, n, s.nullCov, s.nullCov + n, len, base, out, room);
    HIP_TRY(hipGetLastError());
    launches += n > 0 ? 2 : 1;
  }
  HIP_TRY(hipEventRecord(s.ev[2], stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (timing) {
    float t = 0;
    (void)hipEventElapsedTime(&t, s.ev[0], s.ev[2]);
    *timing = inner;
    timing->kernel_ms = kernel_ms + t; timing->total_ms = total_ms + t; timing->n_launches = launches;
  }
  if (s.nullRowsCap * sizeof(int64_t) > kKeepStagingBytes) { (void)hipFree(s.nullRows); s.nullRows = nullptr; s.nullRowsCap = 0; }
  if (s.nullCovCap * sizeof(uint64_t) > kKeepStagingBytes) { (void)hipFree(s.nullCov); s.nullCov = nullptr; s.nullCovCap = 0; }
  if (d_out && total > cap) return fail(CXG_E_CAPACITY, "output capacity too small");
  return CXG_OK;
}

// ---- FindAllSubmatch of a nullable pattern (round 5; meta/findall.go:390-447) ------------------------------------------------------------
// Rows of FindAllIndex (scanNullable: the non-empty variant's rows + the empty matches, Go's skip rule) widened to 2 x groups, then the
// backtracking capture pass over the pattern's own NFA for EVERY row: anchored at the row's start, accepting at its end — for an
// empty row the top-priority empty path, which decides the groups that take part (`(a*)(b)?` at an empty match: group 1 = (p, p),
// group 2 unset).
__global__ void k_null_sub_expand(const int64_t* spans, uint64_t n, uint32_t width, int64_t* out) {
  const uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // one thread per pair of slots
  const uint32_t pairs = width >> 1;
  const uint64_t i = t / pairs;
  const uint32_t k = static_cast<uint32_t>(t % pairs) * 2u;
  if (i >= n) return;
  if (k == 0) cxgdev::store_pair_nt(out + i * width, spans[2 * i], spans[2 * i + 1]);
  else cxgdev::store_pair_nt(out + i * width + k, -1, -1);
}
// The reference's own quirk, kept: a search that STARTS at the end of the haystack answers an empty match with every group unset
// (nfa/pikevm.go:2201-2212: buildCapturesFromSlots(nil, at, at)), and for a nullable pattern the empty match at len is always found by a
// search that starts there.  Only the last row can be that match.
__global__ void k_null_sub_eoi(int64_t* out, uint64_t n, uint32_t width, int64_t end_abs) {
  int64_t* row = out + (n - 1) * width;
  if (row[0] == end_abs && row[1] == end_abs) for (uint32_t k = 2 + threadIdx.x; k < width; k += blockDim.x) row[k] = -1;
}
int scanNullableSubmatch(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out, uint64_t cap,
                         uint64_t* n_out, void* user_stream, cxg_timing* timing, int row_width) {
  if (n_out) *n_out = 0;
  if (!d_out) return scanNullable(p, d_hay, len, base, limit, nullptr, 0, n_out, user_stream, timing);   // a row per match
  if (reinterpret_cast<uintptr_t>(d_out) & 15u) return fail(CXG_E_INVALID, "device output must be 16-byte aligned");
  Scratch* sp;
  if (int rc = getScratch(&sp)) return rc;
  Scratch& s = *sp;
  hipStream_t stream = user_stream ? static_cast<hipStream_t>(user_stream) : s.stream;
  cxg_timing t0;
  std::memset(&t0, 0, sizeof t0);
  float kernel_ms = 0, total_ms = 0;
  uint32_t launches = 0;
  uint64_t n = 0;
  if (int rc = scanNullable(p, d_hay, len, base, limit, nullptr, 0, &n, user_stream, &t0)) return rc;   // the count sizes the span array
  kernel_ms += t0.kernel_ms; total_ms += t0.total_ms; launches += t0.n_launches;
  if (n_out) *n_out = n;
  if (n > cap) return fail(CXG_E_CAPACITY, "output capacity too small");
  if (n == 0) { if (timing) { *timing = t0; } return CXG_OK; }
  if (2 * n + 2 > s.offSpansCap) {
    if (s.offSpans) HIP_TRY(hipFree(s.offSpans));
    s.offSpans = nullptr; s.offSpansCap = 0;
    const uint64_t c = 2 * n + n / 2 + 1024;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.offSpans), c * sizeof(int64_t)));
    s.offSpansCap = c;
  }
  uint64_t n2 = 0;
  if (int rc = scanNullable(p, d_hay, len, base, limit, s.offSpans, n, &n2, user_stream, &t0)) return rc;
  if (n2 != n) return fail(CXG_E_INTERNAL, "nullable captures: the rerun for rows disagrees with the count");
  kernel_ms += t0.kernel_ms; total_ms += t0.total_ms; launches += t0.n_launches;
  const uint8_t* d_cap = nullptr;
  if (int rc = deviceCopy(p->capBlob, &const_cast<cxg_program*>(p)->devCap[t_device], &d_cap)) return rc;
  if (!s.bothFirst) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.bothFirst), 16));
  uint32_t* d_err = reinterpret_cast<uint32_t*>(s.bothFirst);
  OrderGate orderGate(g_path[s.device < 0 ? 0 : s.device], stream);
  HIP_TRY(hipMemsetAsync(d_err, 0, 8, stream));
  HIP_TRY(hipEventRecord(s.ev[0], stream));
  int64_t* out = static_cast<int64_t*>(d_out);
  const uint32_t width = static_cast<uint32_t>(row_width);
  {
    const uint64_t threads = n * (width / 2);
    hipLaunchKernelGGL(k_null_sub_expand, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, stream, s.offSpans, n, width, out);
  }
  {
    const unsigned blk = 64, grd = static_cast<unsigned>(std::min<uint64_t>((n + blk - 1) / blk, 64));
    const size_t need = static_cast<size_t>(grd) * blk * (cxgdev::kBtVisitedWords * 4ull + cxgdev::kBtStackEntries * 8ull);
    if (s.btCap < need) {
      if (s.bt) HIP_TRY(hipFree(s.bt));
      s.bt = nullptr; s.btCap = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.bt), need));
      s.btCap = need;
    }
    const uint32_t img = reinterpret_cast<const cxgdev::BtHeader*>(p->capBlob.data())->total_bytes;
    const uint32_t img_lds = img <= 16384u ? ((img + 3u) & ~3u) : 0u;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned g1 = static_cast<unsigned>(std::min<uint64_t>((n + 255) / 256, static_cast<uint64_t>(cus) * 2u));
    const uint8_t* hay = static_cast<const uint8_t*>(d_hay);
    hipLaunchKernelGGL(k_captures_bt_lds<false>, dim3(g1), dim3(256), img_lds, stream, hay, base, len, out, n, width, d_cap, img_lds, d_err);
    hipLaunchKernelGGL(k_captures_bt<false>, dim3(grd), dim3(blk), 0, stream, hay, base, len, out, n, width, d_cap, s.bt, d_err);
    hipLaunchKernelGGL(k_null_sub_eoi, dim3(1), dim3(64), 0, stream, out, n, width, base + static_cast<int64_t>(len));
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipEventRecord(s.ev[2], stream));
  uint32_t err = 0;
  HIP_TRY(hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  float t = 0;
  (void)hipEventElapsedTime(&t, s.ev[0], s.ev[2]);
  kernel_ms += t; total_ms += t; launches += 3;
  if (timing) { *timing = t0; timing->kernel_ms = kernel_ms; timing->total_ms = total_ms; timing->n_launches = launches; }
  if (s.offSpansCap * sizeof(int64_t) > kKeepStagingBytes) { (void)hipFree(s.offSpans); s.offSpans = nullptr; s.offSpansCap = 0; }
  if (err & cxgdev::kErrSerialLimit) return fail(CXG_E_INPUT, "nullable captures: a match too long for the backtracking pass's budget (65 536 / NFA states bytes)");
  if (err) return fail(CXG_E_INTERNAL, "nullable captures: the backtracking pass found no path for a row (flag " + std::to_string(err) + ")");
  return CXG_OK;
}

// ---- offset captures ---------------------------------------------------------------------------------------------------------
struct OffCapsArg { uint8_t src[32]; int32_t delta[32]; };
__global__ void k_caps_offsets(const int64_t* spans, uint64_t n, uint32_t width, OffCapsArg oc, int64_t* out) {
  const uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // one thread per PAIR of slots: 16-byte stores
  const uint32_t pairs = width >> 1;
  const uint64_t i = t / pairs;
  const uint32_t k = static_cast<uint32_t>(t % pairs) * 2u;
  if (i >= n) return;
  const int64_t s = spans[2 * i], e = spans[2 * i + 1];
  cxgdev::store_pair_nt(out + i * width + k, (oc.src[k] ? e : s) + oc.delta[k], (oc.src[k + 1] ? e : s) + oc.delta[k + 1]);
}
int scanOffsetCaps(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out, uint64_t cap,
                   uint64_t* n_out, void* user_stream, cxg_timing* timing, int row_width) {
  if (n_out) *n_out = 0;
  if (!d_out) return scanDevice(p, d_hay, len, base, limit, nullptr, 0, n_out, user_stream, timing, 2);   // a row per span
  if (reinterpret_cast<uintptr_t>(d_out) & 15u) return fail(CXG_E_INVALID, "device output must be 16-byte aligned");
  Scratch* sp;
  if (int rc = getScratch(&sp)) return rc;
  Scratch& s = *sp;
  hipStream_t stream = user_stream ? static_cast<hipStream_t>(user_stream) : s.stream;
  uint64_t want = cap;
  if (limit > 0 && static_cast<uint64_t>(limit) < want) want = static_cast<uint64_t>(limit);
  cxg_timing t0;
  std::memset(&t0, 0, sizeof t0);
  float kernel_ms = 0, total_ms = 0;
  uint32_t launches = 0;
  if (want * 16u > (256ull << 20)) {                               // a generous cap: size the spans by the count
    uint64_t n = 0;
    if (int rc = scanDevice(p, d_hay, len, base, limit, nullptr, 0, &n, user_stream, &t0, 2)) return rc;
    kernel_ms += t0.kernel_ms; total_ms += t0.total_ms; launches += t0.n_launches;
    if (n > cap) { if (n_out) *n_out = n; return fail(CXG_E_CAPACITY, "output capacity too small"); }
    want = n;
  }
  if (2 * want + 2 > s.offSpansCap) {
    if (s.offSpans) HIP_TRY(hipFree(s.offSpans));
    s.offSpans = nullptr; s.offSpansCap = 0;
    const uint64_t c = 2 * want + want / 2 + 1024;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.offSpans), c * sizeof(int64_t)));
    s.offSpansCap = c;
  }
  uint64_t n = 0;
  int rc = scanDevice(p, d_hay, len, base, limit, s.offSpans, want, &n, user_stream, &t0, 2);
  kernel_ms += t0.kernel_ms; total_ms += t0.total_ms; launches += t0.n_launches;
  if (n_out) *n_out = n;
  if (rc != CXG_OK) return rc;
  if (n > cap) return fail(CXG_E_CAPACITY, "output capacity too small");
  if (n) {
    OffCapsArg oc;
    std::memcpy(oc.src, p->offSrc, sizeof oc.src);
    std::memcpy(oc.delta, p->offDelta, sizeof oc.delta);
    const uint64_t threads = n * static_cast<uint64_t>(row_width / 2);
    OrderGate orderGate(g_path[s.device < 0 ? 0 : s.device], stream);
    HIP_TRY(hipEventRecord(s.ev[0], stream));
    hipLaunchKernelGGL(k_caps_offsets, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, stream, s.offSpans, n, static_cast<uint32_t>(row_width), oc, static_cast<int64_t*>(d_out));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s.ev[2], stream));
    HIP_TRY(hipStreamSynchronize(stream));
    float t = 0;
    (void)hipEventElapsedTime(&t, s.ev[0], s.ev[2]);
    kernel_ms += t; total_ms += t; launches++;
  }
  if (timing) { *timing = t0; timing->kernel_ms = kernel_ms; timing->total_ms = total_ms; timing->n_launches = launches; }
  if (s.offSpansCap * sizeof(int64_t) > kKeepStagingBytes) { (void)hipFree(s.offSpans); s.offSpans = nullptr; s.offSpansCap = 0; }
  return CXG_OK;
}

uint64_t tilesFor(uint32_t kind, uint64_t len) {
  (void)kind;
  return (len + cxgdev::kTile - 1) / cxgdev::kTile;
}

__global__ void k_fill_synth(uint8_t* dst, uint64_t npages, uint32_t config, uint64_t seed, uint64_t first_page) {
  const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
  if (i >= npages) return;
  cxgsynth::page(config, seed, first_page + i, dst + i * cxgsynth::kPage);
}

// Host-memory haystack (what the cgo shim passes): H2D copy into the call's scratch buffer, the device scan,
// D2H copy of the rows.  No CPU compute path exists in this library.
constexpr uint64_t kZeroCopyHay = 256ull << 10;     // bytes of haystack served from pinned host memory
constexpr uint64_t kZeroCopyVals = 128ull << 10;    // int64 values of rows written to pinned host memory (1 MiB)

int scanHostBuffer(const cxg_program* p, const uint8_t* hay, uint64_t len, int64_t limit, int64_t* rows, uint64_t cap,
             uint64_t* n_out, int width) {
  if (!p) return fail(CXG_E_INVALID, "null program");
  if (width > 2 ? !(p->subSupported || (p->offCapsOn && p->supported)) : !p->supported)   // (the predicate of cxg_program_submatch_supported)
    return fail(CXG_E_UNSUPPORTED, width > 2 ? p->subWhyNot : (p->whyNot.empty() ? "unsupported program" : p->whyNot));
  if (n_out) *n_out = 0;
  if (limit == 0 || (len == 0 && !p->nullable)) return CXG_OK;   // (a nullable pattern matches the empty haystack once, captures included)
  Scratch* sp;
  if (int rc = getScratch(&sp)) return rc;
  Scratch& s = *sp;
  // Small haystacks: two hipMemcpy calls cost more than the scan.  Stage the bytes in pinned host memory with a plain
  // memcpy, let the kernels read them over PCIe and write the rows into pinned host memory: one launch + one sync.
  static const bool zeroCopyOk = getenv("CXG_NO_ZERO_COPY") == nullptr;
  if (zeroCopyOk && len <= kZeroCopyHay) {
    if (!s.pinHay) {
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.pinHay), kZeroCopyHay + 4096, hipHostMallocDefault));
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.pinOut), kZeroCopyVals * sizeof(int64_t), hipHostMallocDefault));
    }
    std::memcpy(s.pinHay, hay, len);
    std::memset(s.pinHay + len, 0, 64);
    uint64_t want = rows ? cap : 0;
    if (limit > 0 && static_cast<uint64_t>(limit) < want) want = static_cast<uint64_t>(limit);
    if (want > kZeroCopyVals / static_cast<uint64_t>(width)) want = kZeroCopyVals / static_cast<uint64_t>(width);
    uint64_t n = 0;
    const int rc = scanDevice(p, s.pinHay, len, 0, limit, rows ? s.pinOut : nullptr, want, &n, nullptr, nullptr, width);
    if (rc == CXG_OK) {
      if (n_out) *n_out = n;
      if (rows && n) std::memcpy(rows, s.pinOut, n * width * sizeof(int64_t));
      return CXG_OK;
    }
    if (rc != CXG_E_CAPACITY || want >= cap) { if (n_out) *n_out = n; return rc; }
    // more rows than the pinned array holds and the caller has room for them: the copying path below
  }
  if (len + 64 > s.hayCap) {
    if (s.hay) HIP_TRY(hipFree(s.hay));
    s.hay = nullptr; s.hayCap = 0;
    uint64_t c = len + len / 8 + 4096;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.hay), c));
    s.hayCap = c;
  }
  HIP_TRY(hipMemcpyAsync(s.hay, hay, len, hipMemcpyHostToDevice, s.stream));
  uint64_t want = rows ? cap : 0;
  if (limit > 0 && static_cast<uint64_t>(limit) < want) want = static_cast<uint64_t>(limit);
  if (want * width * 8 > (64ull << 20)) {   // large cap: count first, then size the staging exactly
    uint64_t n = 0;
    if (int rc = scanDevice(p, s.hay, len, 0, limit, nullptr, 0, &n, nullptr, nullptr, width)) return rc;
    if (n > cap) { if (n_out) *n_out = n; return fail(CXG_E_CAPACITY, "output capacity too small"); }
    want = n;
  }
  if (want * width > s.outCap) {
    if (s.out) HIP_TRY(hipFree(s.out));
    s.out = nullptr; s.outCap = 0;
    uint64_t c = want * width + 1024;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.out), c * sizeof(int64_t)));
    s.outCap = c;
  }
  uint64_t n = 0;
  int rc = scanDevice(p, s.hay, len, 0, limit, rows ? s.out : nullptr, want, &n, nullptr, nullptr, width);
  if (n_out) *n_out = n;
  if (rc == CXG_OK && rows && n) {
    const hipError_t ce = hipMemcpy(rows, s.out, n * width * sizeof(int64_t), hipMemcpyDeviceToHost);
    if (ce != hipSuccess) rc = failHip(ce, "hipMemcpy(rows)");
  }
  // a thread keeps at most kKeepStagingBytes of HBM staging between calls
  if (s.hayCap > kKeepStagingBytes) { (void)hipFree(s.hay); s.hay = nullptr; s.hayCap = 0; }
  if (s.outCap * sizeof(int64_t) > kKeepStagingBytes) { (void)hipFree(s.out); s.out = nullptr; s.outCap = 0; }
  return rc;
}

}  // namespace

struct cxg_buffer {
  int device = 0;
  uint8_t* d = nullptr;
  uint64_t len = 0;
};

extern "C" {

const char* cxg_last_error(void) { return t_err.c_str(); }
const char* cxg_version(void) { return "coregex_hip 0.2 (gfx950)"; }
int cxg_abi_version(void) { return CXG_ABI_VERSION; }
size_t cxg_timing_size(void) { return sizeof(cxg_timing); }
int cxg_path_reset(int device) {
  if (device < 0 || device >= 16) return fail(CXG_E_INVALID, "bad device index");
  PathState& ps = g_path[device];
  for (PathMode* m : {&ps.staticGroups, &ps.persistent, &ps.delim}) { m->penalty.store(0); m->backoff.store(8); }
  return CXG_OK;
}
int cxg_debug_demote(int device, int mode) {
  if (device < 0 || device >= 16 || mode < 0 || mode > 2) return fail(CXG_E_INVALID, "bad argument");
  PathState& ps = g_path[device];
  (mode == 0 ? ps.staticGroups : mode == 1 ? ps.persistent : ps.delim).demote();
  return CXG_OK;
}
int cxg_path_state(int device, cxg_path_state_t* out) {
  if (!out || device < 0 || device >= 16) return fail(CXG_E_INVALID, "bad argument");
  const PathState& ps = g_path[device];
  out->static_penalty = ps.staticGroups.penalty.load(); out->static_hits = ps.staticGroups.hits.load();
  out->persistent_penalty = ps.persistent.penalty.load(); out->persistent_hits = ps.persistent.hits.load();
  out->delim_penalty = ps.delim.penalty.load(); out->delim_hits = ps.delim.hits.load();
  out->order_waiters = ps.orderWaiters.load();
  out->reserved = 0;
  return CXG_OK;
}
int cxg_device_count(void) { return deviceCount(); }
int cxg_set_device(int device) {
  if (device < 0 || device >= 16) return fail(CXG_E_INVALID, "bad device index");
  t_device = device;
  return CXG_OK;
}

int cxg_device_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes) {
  if (device < 0 || device >= deviceCount()) return fail(CXG_E_NO_GPU, "no such device");
  int before = 0;
  (void)hipGetDevice(&before);
  HIP_TRY(hipSetDevice(device));
  size_t f = 0, t = 0;
  const hipError_t e = hipMemGetInfo(&f, &t);
  (void)hipSetDevice(before);
  if (e != hipSuccess) return failHip(e, "hipMemGetInfo");
  if (free_bytes) *free_bytes = f;
  if (total_bytes) *total_bytes = t;
  return CXG_OK;
}

void cxg_thread_release(void) {
  for (auto& x : t_scratch) x.release();
}

const char* cxg_kernel_name(int k) {
  switch (k) {
    case CXG_K_DFA_TABLE: return "k_scan_dfa";
    case CXG_K_DIGIT_FLAT: return "k_scan_digit_flat";
    case CXG_K_CHAIN_WAVE: return "k_scan_chain_wave";
    case CXG_K_FIELDS_WAVE: return "k_scan_fields_wave";
    case CXG_K_TRIO_WAVE: return "k_scan_trio_wave";
    case CXG_K_FIELDS_PERS: return "k_scan_fields_pers";
    case CXG_K_DELIM_WAVE: return "k_scan_delim_wave";
    case CXG_K_LITERAL_PERS: return "k_scan_fields_pers<LIT>";
    case CXG_K_TRIO_PERS: return "k_scan_fields_pers<TRIO>";
    case CXG_K_TEDDY_WAVE: return "k_scan_teddy_wave";
    case CXG_K_CHARCLASS_WAVE: return "k_scan_charclass_wave";
    case CXG_K_PREFIX_WAVE: return "k_scan_teddy_wave<VERIFY>";
    case CXG_K_FSM: return "k_scan_fsm";
    case CXG_K_FSM_DIRECT: return "k_scan_fsml<direct>";
    case CXG_K_FSM_LEAN: return "k_scan_fsml";
    case CXG_K_TEDDY_TABLE: return "k_scan_teddy";
    case CXG_K_CHARCLASS_TABLE: return "k_scan_charclass";
    default: return "none";
  }
}

const char* cxg_strategy_name(int s) {
  static const char* n[] = {"UseNFA", "UseDFA", "UseBoth", "UseReverseAnchored", "UseReverseSuffix", "UseOnePass",
                            "UseReverseInner", "UseBoundedBacktracker", "UseTeddy", "UseReverseSuffixSet",
                            "UseCharClassSearcher", "UseCompositeSearcher", "UseBranchDispatch", "UseDigitPrefilter",
                            "UseAhoCorasick", "UseAnchoredLiteral", "UseMultilineReverseSuffix"};
  return (s >= 0 && s < 17) ? n[s] : "?";
}

int cxg_compile(const char* pattern, size_t len, cxg_program** out) {
  if (!pattern || !out) return fail(CXG_E_INVALID, "null argument");
  *out = nullptr;
  try {
    cxg::Ast ast = cxg::parsePattern(std::string(pattern, len));
    auto* p = new cxg_program();
    {
      bool exact = true;
      const int as = cxg::textAnchorStrategy(ast, exact);             // anchored at the text's start or end: engines without a device kernel
      if (as >= 0) {
        p->supported = false; p->strategy = as; p->ngroups = ast.ncap + 1;
        p->whyNot = std::string("strategy ") + cxg_strategy_name(as) + (exact ? "" : " (or UseAnchoredLiteral / UseBranchDispatch)") +
                    " has no device kernel: the pattern is anchored at the " + (as == CXG_USE_REVERSE_ANCHORED ? "end" : "start") + " of the text";
        *out = p;
        return CXG_OK;
      }
    }
    try {
      p->nfa = cxg::buildNfa(ast);
    } catch (const cxg::FrontendError& e) {
      if (e.code != CXG_E_UNSUPPORTED) { delete p; return fail(e.code, e.msg); }
      p->supported = false; p->whyNot = e.msg; p->strategy = CXG_USE_NFA; p->ngroups = ast.ncap + 1;
      *out = p;
      return CXG_OK;
    }
    cxg::Plan plan = cxg::selectStrategy(ast, p->nfa);
    p->ngroups = static_cast<int>(p->nfa.captureCount);
    p->nfaStates = static_cast<int>(p->nfa.states.size());
    cxg_nfa view = p->nfa.view();
    switch (plan.strategy) {
      case CXG_USE_CHARCLASS_SEARCHER: cxg::buildProgramFromCharClass(p, plan.membership, 1); break;
      case CXG_USE_TEDDY: {
        if (plan.lineStart && !plan.lineStartAll) {
          p->supported = false;
          p->whyNot = "(?m)^ on some alternatives only: the reference applies its line-start check to every literal candidate (prefilter.WrapLineAnchor)";
          break;
        }
        if (plan.lineStart) { cxg::buildProgramFromNfa(p, view, CXG_USE_TEDDY, 0); break; }   // (?m)^ + literals: the pattern's transducer
        std::vector<std::vector<uint8_t>> lits;
        for (auto& l : plan.prefixes) lits.push_back(l.bytes);
        cxg::buildProgramFromLiterals(p, lits);
        break;
      }
      default: cxg::buildProgramFromNfa(p, view, plan.strategy, plan.flags); break;
    }
    p->strategy = plan.strategy;
    p->flags = plan.flags;
    p->ngroups = static_cast<int>(p->nfa.captureCount);
    p->nfaStates = static_cast<int>(p->nfa.states.size());
    if (!plan.confident && p->supported) {
      p->supported = false;
      p->whyNot = plan.why.empty() ? "the reference may route this pattern to a reverse-search strategy outside the device subset" : plan.why;
    }
    if (p->ngroups > 1) cxg::buildSubmatchProgram(p, view, plan.strategy);   // FindAllSubmatchIndex path (spans + one-pass captures)
    if (p->ngroups > 1) cxg::deriveOffsetCaps(p, view);
    if (p->supported && p->ngroups == 1) {                     // bounded repetition (`\d{1,3}\.\d{1,3}`...) on the chain kernel
      static const bool noBounded = getenv("CXG_NO_BOUNDED_CHAIN") != nullptr;
      cxg::Ast sur;
      std::vector<std::pair<int, int>> bounds;
      if (!noBounded && cxg::boundedSurrogate(ast, sur, bounds)) {
        try {
          cxg::HostNfa sn = cxg::buildNfa(sur);
          cxg::attachBoundedChain(p, sn.view(), bounds);
        } catch (const cxg::FrontendError&) {
        }
      }
    }
    *out = p;
    return CXG_OK;
  } catch (const cxg::FrontendError& e) {
    return fail(e.code, e.msg);
  } catch (const std::exception& e) {
    return fail(CXG_E_INTERNAL, e.what());
  }
}

// The three constructors a cgo shim calls.  Foreign data: every index of the NFA is validated
// (CXG_E_INVALID + message), nothing thrown crosses the C ABI, and *out is only set on CXG_OK.  A program outside the
// device subset is still CXG_OK with cxg_program_supported() == 0 (the reason in cxg_last_error), like cxg_compile.
int cxg_program_from_nfa(const cxg_nfa* nfa, int strategy, uint32_t flags, cxg_program** out) {
  if (!nfa || !out) return fail(CXG_E_INVALID, "null argument");
  *out = nullptr;
  if (strategy < 0 || strategy > CXG_USE_MULTILINE_REVERSE_SUFFIX) return fail(CXG_E_INVALID, "strategy outside meta.Strategy (0..16)");
  if (flags & ~(CXG_FLAG_DIGIT_RUN_SKIP_SAFE | CXG_FLAG_HAS_REVERSE_DFA | CXG_FLAG_HAS_PREFILTER)) return fail(CXG_E_INVALID, "unknown flag bits");
  cxg_program* p = nullptr;
  try {
    std::string why;
    if (nfa->states)
      for (uint32_t i = 0; i < nfa->n_states; i++)
        if (nfa->states[i].kind == CXG_NFA_RUNE_ANY || nfa->states[i].kind == CXG_NFA_RUNE_ANY_NOT_NL)
          return fail(CXG_E_UNSUPPORTED, "state " + std::to_string(i) + ": nfa.StateRuneAny / StateRuneAnyNotNL (the PikeVM's rune NFA) has no device form; pass Engine.nfa");
    if (!cxg::validateNfa(*nfa, why)) return fail(CXG_E_INVALID, why);
    p = new cxg_program();
    cxg::buildProgramFromNfa(p, *nfa, strategy, flags);
    // FindAllSubmatch hook (meta/findall.go:390): spans + capture table, same call as cxg_compile makes
    if (nfa->capture_count > 1) cxg::buildSubmatchProgram(p, *nfa, strategy);
    if (nfa->capture_count > 1) cxg::deriveOffsetCaps(p, *nfa);
    else p->subWhyNot = "pattern has no capture groups (cxg_find_all_submatch then returns the spans)";
    if (!p->supported) t_err = p->whyNot;
    *out = p;
    return CXG_OK;
  } catch (const cxg::BuildError& e) {
    delete p;
    return fail(e.code, e.msg);
  } catch (const std::exception& e) {
    delete p;
    return fail(CXG_E_INTERNAL, e.what());
  } catch (...) {
    delete p;
    return fail(CXG_E_INTERNAL, "unknown exception");
  }
}

int cxg_program_from_literals(const uint8_t* const* lits, const uint32_t* lens, uint32_t n, cxg_program** out) {
  if (!lits || !lens || !out) return fail(CXG_E_INVALID, "null argument");
  *out = nullptr;
  if (n == 0 || n > 4096) return fail(CXG_E_INVALID, "literal count must be 1..4096");
  cxg_program* p = nullptr;
  try {
    std::vector<std::vector<uint8_t>> v;
    for (uint32_t i = 0; i < n; i++) {
      if (!lits[i] && lens[i]) return fail(CXG_E_INVALID, "null literal pointer");
      if (lens[i] > (1u << 16)) return fail(CXG_E_INVALID, "literal longer than 64 KiB");
      v.emplace_back(lits[i], lits[i] + lens[i]);
    }
    p = new cxg_program();
    cxg::buildProgramFromLiterals(p, v);
    p->subWhyNot = "literal set has no capture groups";
    if (!p->supported) t_err = p->whyNot;
    *out = p;
    return CXG_OK;
  } catch (const std::exception& e) {
    delete p;
    return fail(CXG_E_INTERNAL, e.what());
  } catch (...) {
    delete p;
    return fail(CXG_E_INTERNAL, "unknown exception");
  }
}

int cxg_program_from_charclass(const uint8_t membership[256], uint32_t min_match, cxg_program** out) {
  if (!membership || !out) return fail(CXG_E_INVALID, "null argument");
  *out = nullptr;
  if (min_match == 0) return fail(CXG_E_INVALID, "min_match must be >= 1 (CharClassSearcher.minMatch, nfa/charclass_searcher.go:26)");
  cxg_program* p = nullptr;
  try {
    bool any = false;
    for (int b = 0; b < 256; b++) any = any || membership[b] != 0;
    if (!any) return fail(CXG_E_INVALID, "empty membership table");
    p = new cxg_program();
    cxg::buildProgramFromCharClass(p, membership, min_match);
    p->subWhyNot = "char-class searcher has no capture groups";
    if (!p->supported) t_err = p->whyNot;
    *out = p;
    return CXG_OK;
  } catch (const std::exception& e) {
    delete p;
    return fail(CXG_E_INTERNAL, e.what());
  } catch (...) {
    delete p;
    return fail(CXG_E_INTERNAL, "unknown exception");
  }
}

void cxg_program_destroy(cxg_program* p) {
  if (!p) return;
  for (int d = 0; d < 16; d++) {
    if (p->dev[d] || p->devSub[d] || p->devCap[d] || p->devFsm[d] || p->devSubFsm[d]) (void)hipSetDevice(d);
    if (p->dev[d]) (void)hipFree(p->dev[d]);
    if (p->devSub[d]) (void)hipFree(p->devSub[d]);
    if (p->devCap[d]) (void)hipFree(p->devCap[d]);
    if (p->devFsm[d]) (void)hipFree(p->devFsm[d]);
    if (p->devSubFsm[d]) (void)hipFree(p->devSubFsm[d]);
  }
  delete p;
}

int cxg_program_strategy(const cxg_program* p) { return p ? p->strategy : -1; }
uint32_t cxg_program_flags(const cxg_program* p) { return p ? p->flags : 0u; }
int cxg_program_num_groups(const cxg_program* p) { return p ? p->ngroups : 0; }
int cxg_program_nfa_states(const cxg_program* p) { return p ? p->nfaStates : -1; }
int cxg_program_dfa_states(const cxg_program* p) { return p ? static_cast<int>(p->fwd.nstates) : 0; }
int cxg_program_delimiters(const cxg_program* p, int* open_byte, int* close_byte, int* plus) {
  if (!p || !p->supported || p->delim[3] == 0u) return 0;
  if (open_byte) *open_byte = static_cast<int>(p->delim[0]);
  if (close_byte) *close_byte = static_cast<int>(p->delim[1]);
  if (plus) *plus = static_cast<int>(p->delim[2]);
  return 1;
}
int cxg_program_nullable(const cxg_program* p) { return !p || !p->nullable ? 0 : (p->nullableOnlyEmpty ? 2 : 1); }
int cxg_program_supported(const cxg_program* p) {
  if (p && !p->supported) t_err = p->whyNot;
  return p && p->supported ? 1 : 0;
}
int cxg_program_blob(const cxg_program* p, const void** data, size_t* len) {
  if (!p || !data || !len) return fail(CXG_E_INVALID, "null argument");
  if (!p->supported) return fail(CXG_E_UNSUPPORTED, p->whyNot);
  *data = p->blob.data();
  *len = p->blob.size();
  return CXG_OK;
}
int cxg_program_fsm_image(const cxg_program* p, int submatch, const void** data, size_t* len) {
  if (!p || !data || !len) return fail(CXG_E_INVALID, "null argument");
  const std::vector<uint8_t>& b = submatch ? p->subFsmBlob : p->fsmBlob;
  if (b.empty()) return fail(CXG_E_UNSUPPORTED, p->fsmWhyNot.empty() ? "program has no FindAll transducer image" : p->fsmWhyNot);
  *data = b.data();
  *len = b.size();
  return CXG_OK;
}
int cxg_program_submatch_blobs(const cxg_program* p, const void** sb, size_t* sl, const void** cb, size_t* cl) {
  if (!p || !sb || !sl || !cb || !cl) return fail(CXG_E_INVALID, "null argument");
  if (!p->subSupported) return fail(CXG_E_UNSUPPORTED, p->subWhyNot);
  *sb = p->subBlob.data(); *sl = p->subBlob.size(); *cb = p->capBlob.data(); *cl = p->capBlob.size();
  return CXG_OK;
}
int cxg_program_chain_captures(const cxg_program* p, uint8_t out[40]) {
  if (!p || !out || !p->subSupported || !p->chainCaps[0]) return 0;
  std::memcpy(out, p->chainCaps, 40);
  return 1;
}
int cxg_program_chain_bounds(const cxg_program* p, uint8_t out[40]) {
  if (!p || !out || !p->supported || p->chainBounds[0] != 2) return 0;
  std::memcpy(out, p->chainBounds, 40);
  return 1;
}
int cxg_program_submatch_supported(const cxg_program* p) {
  if (p && !p->subSupported && !(p->offCapsOn && p->supported)) t_err = p->subWhyNot;
  return p && (p->subSupported || (p->offCapsOn && p->supported)) ? 1 : 0;
}
int cxg_program_offset_captures(const cxg_program* p, int* src, int* delta, int max_slots) {
  if (!p || !p->offCapsOn) return 0;
  const int n = 2 * p->ngroups;
  for (int k = 0; k < n && k < max_slots; k++) { if (src) src[k] = p->offSrc[k]; if (delta) delta[k] = p->offDelta[k]; }
  return n;
}
int cxg_program_nfa(const cxg_program* p, cxg_nfa* out) {
  if (!p || !out) return fail(CXG_E_INVALID, "null argument");
  if (p->nfa.states.empty()) return fail(CXG_E_INVALID, "program was not built by cxg_compile");
  *out = p->nfa.view();
  return CXG_OK;
}

int cxg_find_all(const cxg_program* p, const uint8_t* hay, uint64_t len, int64_t limit, int64_t* spans, uint64_t cap,
                 uint64_t* n_out) {
  return scanHostBuffer(p, hay, len, limit, spans, cap, n_out, 2);
}
int cxg_count(const cxg_program* p, const uint8_t* hay, uint64_t len, int64_t limit, uint64_t* n_out) {
  return scanHostBuffer(p, hay, len, limit, nullptr, 0, n_out, 2);
}
// Engine.Find (meta/find.go:29) and Engine.IsMatch (meta/ismatch.go:27) of a whole haystack: FindAll with n == 1 — the first match in
// haystack order.  The early stop is FindAll's: the group whose look-back has counted the first row raises the stop word, groups that
// start afterwards publish and leave (block_common.hpp limit_reached_skip), so a haystack with an early match costs the groups that
// were resident, not its length.  An empty match counts (nullable programs go through their merge with the same limit).
int cxg_find(const cxg_program* p, const uint8_t* hay, uint64_t len, int64_t span[2], int* found) {
  if (!span || !found) return fail(CXG_E_INVALID, "null argument");
  uint64_t n = 0;
  int64_t row[2] = {-1, -1};
  *found = 0;
  if (const int rc = scanHostBuffer(p, hay, len, 1, row, 1, &n, 2)) return rc;
  if (n != 0) { *found = 1; span[0] = row[0]; span[1] = row[1]; }
  return CXG_OK;
}
int cxg_is_match(const cxg_program* p, const uint8_t* hay, uint64_t len, int* matched) {
  if (!matched) return fail(CXG_E_INVALID, "null argument");
  uint64_t n = 0;
  *matched = 0;
  if (const int rc = scanHostBuffer(p, hay, len, 1, nullptr, 0, &n, 2)) return rc;
  *matched = n != 0 ? 1 : 0;
  return CXG_OK;
}
int cxg_find_all_submatch(const cxg_program* p, const uint8_t* hay, uint64_t len, int64_t limit, int64_t* slots,
                          uint64_t cap, uint64_t* n_out) {
  if (p && p->ngroups == 1) return scanHostBuffer(p, hay, len, limit, slots, cap, n_out, 2);
  return scanHostBuffer(p, hay, len, limit, slots, cap, n_out, p ? 2 * p->ngroups : 2);
}

int cxg_buffer_alloc(uint64_t len, cxg_buffer** out) {
  if (!out) return fail(CXG_E_INVALID, "null argument");
  Scratch* sp;
  if (int rc = getScratch(&sp)) return rc;
  auto* b = new cxg_buffer();
  b->device = t_device;
  b->len = len;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&b->d), len + 4096);
  if (e != hipSuccess) { delete b; return failHip(e, "hipMalloc"); }
  *out = b;
  return CXG_OK;
}
void cxg_buffer_free(cxg_buffer* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  (void)hipFree(b->d);
  delete b;
}
int cxg_buffer_upload(cxg_buffer* b, uint64_t off, const uint8_t* src, uint64_t len) {
  if (!b || off + len > b->len) return fail(CXG_E_INVALID, "bad range");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipMemcpy(b->d + off, src, len, hipMemcpyHostToDevice));
  return CXG_OK;
}
int cxg_buffer_download(const cxg_buffer* b, uint64_t off, uint8_t* dst, uint64_t len) {
  if (!b || off + len > b->len) return fail(CXG_E_INVALID, "bad range");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipMemcpy(dst, b->d + off, len, hipMemcpyDeviceToHost));
  return CXG_OK;
}
uint64_t cxg_buffer_len(const cxg_buffer* b) { return b ? b->len : 0; }
void* cxg_buffer_device_ptr(const cxg_buffer* b) { return b ? b->d : nullptr; }

int cxg_buffer_fill_synth(cxg_buffer* b, uint32_t config, uint64_t seed, uint64_t first_page) {
  if (!b || b->len % cxgsynth::kPage) return fail(CXG_E_INVALID, "buffer length must be a multiple of 4096");
  HIP_TRY(hipSetDevice(b->device));
  const uint64_t npages = b->len / cxgsynth::kPage;
  if (npages == 0) return CXG_OK;
  const unsigned block = 64;
  const unsigned grid = static_cast<unsigned>((npages + block - 1) / block);
  {
    OrderGate orderGate(g_path[b->device], nullptr);               // (a fill beside another thread's persistent scan would be a foreign kernel to it)
    hipLaunchKernelGGL(k_fill_synth, dim3(grid), dim3(block), 0, nullptr, b->d, npages, config, seed, first_page);
    HIP_TRY(hipGetLastError());
  }
  HIP_TRY(hipDeviceSynchronize());
  return CXG_OK;
}
int cxg_synth_page_host(uint32_t config, uint64_t seed, uint64_t page, uint8_t out[4096]) {
  cxgsynth::page(config, seed, page, out);
  return CXG_OK;
}

int cxg_find_all_device(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out,
                        uint64_t cap, uint64_t* n_out, void* stream, cxg_timing* timing) {
  return scanDevice(p, d_hay, len, base, limit, d_out, cap, n_out, stream, timing, 2);
}
// The same two questions of a device-resident haystack (a shard): the row comes back through 16 bytes of the thread's scratch.
int cxg_find_device(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t span[2], int* found, void* stream) {
  if (!span || !found) return fail(CXG_E_INVALID, "null argument");
  *found = 0;
  Scratch* sp;
  if (int rc = getScratch(&sp)) return rc;
  if (!sp->findRow) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&sp->findRow), 64));
  uint64_t n = 0;
  if (const int rc = scanDevice(p, d_hay, len, base, 1, sp->findRow, 1, &n, stream, nullptr, 2)) return rc;
  if (n != 0) {
    int64_t row[2];
    HIP_TRY(hipMemcpy(row, sp->findRow, 16, hipMemcpyDeviceToHost));
    *found = 1; span[0] = row[0]; span[1] = row[1];
  }
  return CXG_OK;
}
int cxg_is_match_device(const cxg_program* p, const void* d_hay, uint64_t len, int* matched, void* stream) {
  if (!matched) return fail(CXG_E_INVALID, "null argument");
  uint64_t n = 0;
  *matched = 0;
  if (const int rc = scanDevice(p, d_hay, len, 0, 1, nullptr, 0, &n, stream, nullptr, 2)) return rc;
  *matched = n != 0 ? 1 : 0;
  return CXG_OK;
}
// ---- asynchronous device entry (round 5) -------------------------------------------------------------------------------------
// cxg_find_all_device without the stream synchronisation at its end: the first span launch of the call is left in flight and
// the handle is waited for later.  A host that scans many shards / haystacks back to back pays the ~19 us of launch + sync +
// pinned read-back once per batch instead of once per call (bench.py: 1 GiB, 4 300 -> 4 650 GB/s).  What cannot be left pending
// (nullable, UseBoth and offset-capture programs, kernels without epoch-tagged status words) runs to completion inside the
// async call; cxg_wait then only hands the result over.  A launch that asked for another rung of the ladder (match-dense input,
// a watchdog) is rerun synchronously by cxg_wait: the result is always what cxg_find_all_device would have returned.
struct cxg_pending { Scratch* s; int slot; int device; };

int cxg_find_all_device_async(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out,
                              uint64_t cap, void* stream, cxg_pending** out) {
  if (!out) return fail(CXG_E_INVALID, "null argument");
  *out = nullptr;
  if (!p) return fail(CXG_E_INVALID, "null program");
  Scratch* sp;
  if (int rc = getScratch(&sp)) return rc;
  Scratch& s = *sp;
  int slot = -1;
  for (int i = 0; i < Scratch::kAsyncSlots; i++) if (!s.async[i].busy) { slot = i; break; }
  if (slot < 0) return fail(CXG_E_CAPACITY, "16 asynchronous calls of this thread are pending: cxg_wait for one first");
  if (!s.asyncCtl) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s.asyncCtl), Scratch::kAsyncSlots * 16, hipHostMallocDefault));
    std::memset(s.asyncCtl, 0, Scratch::kAsyncSlots * 16);
  }
  Scratch::AsyncSlot& as = s.async[slot];
  for (auto& e : as.ev) if (!e) HIP_TRY(hipEventCreate(&e));
  as.ctl = s.asyncCtl + 2 * slot;
  as.p = p; as.hay = d_hay; as.len = len; as.base = base; as.limit = limit; as.out = d_out; as.cap = cap; as.user_stream = stream;
  as.done = false; as.done_rc = 0; as.done_n = 0;
  std::memset(&as.timing, 0, sizeof as.timing);
  const bool plain = p->supported && !p->nullable && limit != 0 && len != 0;
  int rc;
  uint64_t n = 0;
  if (plain) {
    t_asyncSlot = &as;
    rc = scanDeviceOnce(p, d_hay, len, base, limit, d_out, cap, &n, stream, &as.timing, 2);
    t_asyncSlot = nullptr;
    if (rc == kRcLongMatch) rc = scanDevice(p, d_hay, len, base, limit, d_out, cap, &n, stream, &as.timing, 2);   // (UseBoth restart: the synchronous loop)
  } else {
    rc = scanDevice(p, d_hay, len, base, limit, d_out, cap, &n, stream, &as.timing, 2);
  }
  if (rc != kRcPending) { as.done = true; as.done_rc = rc; as.done_n = n; }
  as.busy = true;
  *out = new cxg_pending{sp, slot, t_device};
  return CXG_OK;
}

int cxg_wait(cxg_pending* h, uint64_t* n_out, cxg_timing* timing) {
  if (!h) return fail(CXG_E_INVALID, "null handle");
  Scratch* sp = nullptr;
  const int dev_before = t_device;
  t_device = h->device;
  const int grc = getScratch(&sp);
  t_device = dev_before;
  if (grc != CXG_OK || sp != h->s) { return fail(CXG_E_THREAD, "cxg_wait must be called on the thread that made the asynchronous call (the handle stays valid there)"); }
  Scratch& s = *sp;
  Scratch::AsyncSlot& as = s.async[h->slot];
  delete h;
  if (!as.busy) return fail(CXG_E_INVALID, "stale handle");
  if (n_out) *n_out = 0;
  int rc;
  if (as.done) {
    rc = as.done_rc;
    if (n_out) *n_out = as.done_n;
    if (timing) *timing = as.timing;
    as.busy = false;
    if (rc != CXG_OK) t_err = "asynchronous call failed when it was made (code " + std::to_string(rc) + ")";
    return rc;
  }
  const hipError_t we = hipEventSynchronize(as.ev[1]);
  const uint64_t total = as.ctl[0];
  const uint32_t err = static_cast<uint32_t>(as.ctl[1]);
  PathState& ps = g_path[s.device];
  --s.asyncInFlight;
  as.busy = false;
  if (we != hipSuccess) return failHip(we, "hipEventSynchronize");
  if (err != 0) {                                                   // another rung of the ladder (or a watchdog): the synchronous call decides and demotes
    t_device = s.device;
    rc = scanDevice(as.p, as.hay, as.len, as.base, as.limit, as.out, as.cap, n_out, as.user_stream, timing, 2);
    t_device = dev_before;
    return rc;
  }
  if (as.mode == 3u) ps.delim.clean(); else if (as.mode == 2u) ps.persistent.clean(); else if (as.mode == 1u) ps.staticGroups.clean();
  if (timing) {
    std::memset(timing, 0, sizeof *timing);
    float k = 0;
    static const bool asyncTiming = getenv("CXG_ASYNC_TIMING") != nullptr;
    if (asyncTiming) (void)hipEventElapsedTime(&k, as.ev[0], as.ev[1]);   // else 0: pending launches carry no start event
    timing->kernel_ms = k; timing->total_ms = k; timing->n_launches = 1; timing->n_ladder = 1; timing->ladder[0] = static_cast<uint8_t>(as.kernelId);
    timing->kernel = as.kernelId; timing->block = cxgdev::kThreads; timing->tiles = as.tiles; timing->grid = static_cast<uint32_t>(as.tiles);
  }
  uint64_t n = total;
  if (as.limit > 0 && n > static_cast<uint64_t>(as.limit)) n = static_cast<uint64_t>(as.limit);
  if (n_out) *n_out = n;
  if (as.out && n > as.cap) return fail(CXG_E_CAPACITY, "output capacity too small");
  return CXG_OK;
}

int cxg_find_all_device_u32(const cxg_program* p, const void* d_hay, uint64_t len, int64_t limit, void* d_out_u32, uint64_t cap,
                            uint64_t* n_out, void* stream, cxg_timing* timing) {
  if (p && (p->nullable || (p->supported && (reinterpret_cast<const cxgdev::BlobHeader*>(p->blob.data())->flags & cxgdev::kFlagBothRestart))))
    return fail(CXG_E_UNSUPPORTED, "compact rows (cxg_find_all_device_u32): not for nullable or UseBoth programs");
  t_u32Rows = true;
  const int rc = scanDevice(p, d_hay, len, 0, limit, d_out_u32, cap, n_out, stream, timing, 2);
  t_u32Rows = false;
  return rc;
}
int cxg_find_all_submatch_device(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit,
                                 void* d_out, uint64_t cap, uint64_t* n_out, void* stream, cxg_timing* timing) {
  return scanDevice(p, d_hay, len, base, limit, d_out, cap, n_out, stream, timing, p ? 2 * p->ngroups : 2);
}

}  // extern "C"
