// capi_host.hip — host haystacks (what the cgo shim passes) and the synthetic corpus fill.
#include "capi_internal.hpp"

namespace cxgapi {

__global__ void k_fill_synth(uint8_t* dst, uint64_t npages, uint32_t config, uint64_t seed, uint64_t first_page) {
  const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
  if (i >= npages) return;
  cxgsynth::page(config, seed, first_page + i, dst + i * cxgsynth::kPage);
}

hipError_t launchFillSynth(uint8_t* dst, uint64_t npages, uint32_t config, uint64_t seed, uint64_t first_page) {
  const unsigned block = 64, grid = static_cast<unsigned>((npages + block - 1) / block);
  hipLaunchKernelGGL(k_fill_synth, dim3(grid), dim3(block), 0, nullptr, dst, npages, config, seed, first_page);
  return hipGetLastError();
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


}  // namespace cxgapi
