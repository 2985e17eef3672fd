// capi_internal.hpp — what the translation units of the C ABI share (round 6: capi.hip was 2 200 lines): error state, launch-mode state per
// device, the per-thread scratch, and the entry points of the pieces — capi_state.hip (device and scratch helpers), capi_captures.hip (the
// capture passes), capi_ladder.hip (one device call: the relaunch ladder of scanDeviceOnce, scanDevice), capi_nullable.hip (nullable programs,
// offset captures), capi_host.hip (host haystacks, corpus fills), capi.hip (the extern "C" functions).  Not installed, not part of the ABI.
#pragma once
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
hipError_t launch_scan_teddy_pair(const ScanArgs& a, uint32_t workgroups, hipStream_t stream);
hipError_t launch_scan_charclass_wave(const ScanArgs& a, hipStream_t stream);
hipError_t launch_scan_fsm(const ScanArgs& a, uint32_t lds_bytes, bool shallow, int look, hipStream_t stream, uint32_t direct_bytes, bool lean, int mode);
}  // namespace cxgdev


namespace cxgapi {

extern thread_local std::string t_err;
extern thread_local int t_device;

inline int fail(int code, const std::string& msg) { t_err = msg; return code; }
inline int failHip(hipError_t e, const char* what) {
  t_err = std::string(what) + ": " + hipGetErrorString(e);
  return CXG_E_DEVICE;
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) return failHip(_e, #expr);    \
  } while (0)

extern std::atomic<bool> g_exiting;   // set by an atexit hook: the HIP runtime may already be gone, leave its memory to the OS

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
extern PathState g_path[16];

int deviceCount();

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
  uint32_t* pairCtr = nullptr; uint32_t pairSeq = 0;     // scan_teddy_pair.hip: two sets of 8 group counters a cache line apart; launch n claims from set n & 1 and zeroes the other
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
      if (pairCtr) (void)hipFree(pairCtr);
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
extern thread_local ScratchSet t_scratch_set;
#define t_scratch t_scratch_set.s
// Staging buffers above this size are returned after the call instead of being kept for the thread's lifetime.
constexpr uint64_t kKeepStagingBytes = 256ull << 20;

hipError_t syncStream(hipStream_t stream);
int getScratch(Scratch** out);
int ensureStatus(Scratch& s, uint64_t ntiles);
int deviceCopy(const std::vector<uint8_t>& host, void** slot, const uint8_t** out);
int deviceBlob(const cxg_program* p, int device, const uint8_t** out);

// (The small kernels outside scanDeviceOnce — merges of nullable programs, capture expansions, corpus fills — are launch sections too:
// OrderGate gate(g_path[device], stream) around their launches.)

uint64_t tilesFor(uint32_t kind, uint64_t len);
int scanNullable(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out, uint64_t cap,
                 uint64_t* n_out, void* user_stream, cxg_timing* timing);
int scanOffsetCaps(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out, uint64_t cap,
                   uint64_t* n_out, void* user_stream, cxg_timing* timing, int row_width);
int scanNullableSubmatch(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out, uint64_t cap,
                         uint64_t* n_out, void* user_stream, cxg_timing* timing, int row_width);
extern thread_local bool t_u32Rows;                                 // cxg_find_all_device_u32 in progress on this thread (ScanArgs::u32_rows)
extern thread_local Scratch::AsyncSlot* t_asyncSlot;               // cxg_find_all_device_async in progress on this thread: leave the first launch pending if it can be
constexpr int kRcPending = -1001;                                  // (internal) scanDeviceOnce left its launch in the slot

// scanDeviceOnce: one search from the haystack's first byte.  kRcLongMatch (internal): a UseBoth program met a match longer
// than its restart span; *n_out = rows of plain leftmost-first iteration, the rows themselves are in d_out when it has room.
constexpr int kRcLongMatch = -1000;

int digitKernelGeneration();
hipError_t launchBtCapturesPlain(unsigned g1, uint32_t img_lds, unsigned grd, unsigned blk, hipStream_t stream, const uint8_t* hay, int64_t base, uint64_t len,
                                 int64_t* out, uint64_t n, uint32_t width, const uint8_t* d_cap, uint8_t* bt, uint32_t* d_err);
int launchCapturePass(const cxg_program* p, Scratch& s, const cxgdev::ScanArgs& a, const uint8_t* d_cap, hipStream_t stream, uint32_t& launches);
int scanDeviceOnce(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out,
                   uint64_t cap, uint64_t* n_out, void* user_stream, cxg_timing* timing, int row_width);
int scanDevice(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit, void* d_out,
               uint64_t cap, uint64_t* n_out, void* user_stream, cxg_timing* timing, int row_width);
int scanHostBuffer(const cxg_program* p, const uint8_t* hay, uint64_t len, int64_t limit, int64_t* rows, uint64_t cap, uint64_t* n_out, int width);
hipError_t launchFillSynth(uint8_t* dst, uint64_t npages, uint32_t config, uint64_t seed, uint64_t first_page);

}  // namespace cxgapi
