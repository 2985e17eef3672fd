// capi_captures.hip — the capture passes behind a span launch (FindAllSubmatchIndex): one-pass table, LDS form, bounded backtracking.
#include "capi_internal.hpp"

namespace cxgapi {

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



// the backtracking pass without assertions over rows that are already on the device (capi_nullable.hip: FindAllSubmatch of a nullable pattern)
hipError_t launchBtCapturesPlain(unsigned g1, uint32_t img_lds, unsigned grd, unsigned blk, hipStream_t stream, const uint8_t* hay, int64_t base, uint64_t len,
                                 int64_t* out, uint64_t n, uint32_t width, const uint8_t* d_cap, uint8_t* bt, uint32_t* d_err) {
  hipLaunchKernelGGL(k_captures_bt_lds<false>, dim3(g1), dim3(256), img_lds, stream, hay, base, len, out, n, width, d_cap, img_lds, d_err);
  hipLaunchKernelGGL(k_captures_bt<false>, dim3(grd), dim3(blk), 0, stream, hay, base, len, out, n, width, d_cap, bt, d_err);
  return hipGetLastError();
}

}  // namespace cxgapi
