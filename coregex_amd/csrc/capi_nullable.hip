// capi_nullable.hip — nullable programs (rows of the non-empty variant + the empty matches), their FindAllSubmatch, and offset captures.
#include "capi_internal.hpp"

namespace cxgapi {

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
    (void)launchBtCapturesPlain(g1, img_lds, grd, blk, stream, hay, base, len, out, n, width, d_cap, s.bt, d_err);   // (capi_captures.hip; errors surface below)
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


}  // namespace cxgapi
