// capi_ladder.hip — one device call: scanDeviceOnce (which kernel serves the program, the launch, and what the error word says comes next —
// the relaunch ladder) and scanDevice (nullable programs, offset captures, the UseBoth restart around it).
#include "capi_internal.hpp"

namespace cxgapi {

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
  a.pair_ctr = nullptr; a.pair_seq = 0; a.pair_nctr = 0; a.pair_nbig = 0; a.pair_n6 = 0; a.pair_n4 = 0;
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
  int fsmMode = p->fsmMode[submatch ? 1 : 0].load(std::memory_order_relaxed);                // transducer kernel: 0, 1 (dense), 2 and 3 (very dense: 1 024 / 2 048 rows per tile)
  static const bool fsmDirectOk = getenv("CXG_FSM_NO_DIRECT") == nullptr;                     // A/B: the class-indexed tables for every machine
  static const bool fsmLeanOk = getenv("CXG_FSM_NO_LEAN") == nullptr;                         // A/B: k_scan_fsm for every machine
  // the lean kernel (scan_fsm.hip k_scan_fsml: shallow machines, entry states that collapse; byte-indexed rows where the image has them)
  // while it serves the program's input
  bool fsmDirect = fsmLeanOk && p->fsmNoDirect[submatch ? 1 : 0].load(std::memory_order_relaxed) == 0;
  bool fsmDirectRan = false, fsmDirectTables = false;
  // literal sets: the pair kernel (scan_teddy_pair.hip, round 6) for the spans of a plain call — FindAll's n lives in the grouped kernels'
  // look-back, and match-dense input (a row buffer overflowed before) or a fallback flag of the pair kernel itself stay on the wave kernel
  static const bool pairOk = getenv("CXG_NO_TEDDY_PAIR") == nullptr;
  // (Until the first two groups of a workgroup were assigned without an atomic, a launch began with 512 atomics queueing on the counters and
  // the wave kernel was ahead below 320 MiB; now the pair kernel is level or ahead from 1 KiB on — r06_c83_pair_sizes_final.txt: 64 KiB 13.6
  // against 23.0 us, 1 MiB 17.4 / 33.7, 64 MiB 43.5 / 48.0, 1 GiB 294 / 405.  CXG_PAIR_MIN_BYTES moves the border for A/B runs.)
  static const uint64_t pairMinBytes = getenv("CXG_PAIR_MIN_BYTES") ? strtoull(getenv("CXG_PAIR_MIN_BYTES"), nullptr, 10) : 0ull;
  if (gen == 7 && pairOk && reinterpret_cast<const cxgdev::TeddyAux*>(p->blob.data() + h->aux_off)->pair_off != 0u && len >= pairMinBytes && limit <= 0 && !denseChain && !profOn && dbgBits == 0 && p->noPair[submatch ? 1 : 0].load(std::memory_order_relaxed) == 0) gen = 12;
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
  if (gen == 12) {                                                 // groups of 8 tiles per wave; the last stretch in smaller ones
    static int pcus = 0;
    if (pcus == 0) { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); pcus = n > 0 ? n : 256; }
    static const int taper = getenv("CXG_PAIR_TAPER") ? atoi(getenv("CXG_PAIR_TAPER")) : 0;   // (1: one group of 6, 4 and 2 tiles per wave and CU behind the big ones — measured: +5 us at 1 GiB, r06_c80_pair_taper.txt; 0: an eighth of a big group per CU in groups of 2)
    const uint64_t rowBytes = static_cast<uint64_t>(cxgdev::kWaveTile) * cxgdev::kPairWaves;   // one tile per wave
    const uint64_t R = (len + rowBytes - 1) / rowBytes;
    const uint64_t P = static_cast<uint64_t>(pcus);
    static const uint64_t tailRows = getenv("CXG_PAIR_TAIL_ROWS") ? static_cast<uint64_t>(atoi(getenv("CXG_PAIR_TAIL_ROWS"))) : 1u;   // (A/B: tile rows per CU left to the groups of 2)
    uint64_t n8 = 0, n6 = 0, n4 = 0;
    if (taper && R >= 20 * P) { n6 = P; n4 = P; n8 = (R - 12 * P) / 8; }          // (behind them: P groups of 2 and what the division left)
    else if (!taper && R >= 9 * P) n8 = (R - tailRows * P) / 8;                 // (a short haystack — less than nine tile rows per CU — is cut into groups of 2 only: more CUs get to work)
    const uint64_t left = R - 8 * n8 - 6 * n6 - 4 * n4;
    a.pair_nbig = static_cast<uint32_t>(n8); a.pair_n6 = static_cast<uint32_t>(n6); a.pair_n4 = static_cast<uint32_t>(n4);
    a.ngroups = n8 + n6 + n4 + (left + 1) / 2;
  }
  a.tiles_per_wave = cxgdev::kTilesPerWave;
  if (((gen == 6 || gen == 7 || gen == 9) && denseChain) || (gen == 10 && fsmMode != 0)) {   // four times the row-buffer room per wave-tile
    a.tiles_per_wave = (gen == 10 && fsmMode >= 2) ? 1u : static_cast<uint32_t>(cxgdev::kDenseTilesPerWave);   // transducer kernel, modes 2 and 3: one tile, 1 024 / 2 048 rows
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
    le = cxgdev::launch_scan_fsm(a, fh->lds_bytes, fh->depth <= 1 && !deepOnly, fh->end_col != 0u ? 2 : (fh->nk > 1 ? 1 : 0), stream, fsmDirectTables ? fh->direct_bytes : 0u, fsmDirectRan, fsmMode);
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
  else if (gen == 12) {                                             // one workgroup per CU, groups claimed (scan_teddy_pair.hip)
    static int cus = 0;
    if (cus == 0) { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); cus = n > 0 ? n : 256; }
    if (!s.pairCtr) {
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.pairCtr), 2 * 8 * cxgdev::kPairCtrStride * sizeof(uint32_t)));
      HIP_TRY(hipMemsetAsync(s.pairCtr, 0, 2 * 8 * cxgdev::kPairCtrStride * sizeof(uint32_t), stream));
      s.pairSeq = 0;
    }
    a.pair_ctr = s.pairCtr; a.pair_seq = ++s.pairSeq;
    // ONE counter: strict ticket order.  With several counters (workgroup b asks counter b & 7 first and STEALS from the others once its own is
    // exhausted) a workgroup can draw a group in front of one it has scanned and not yet resolved — its own look-back then waits for a group it
    // holds itself (seen as [21, 21]: a watchdog rerun, on short haystacks where the first claims outnumber the groups).  In ticket order every
    // workgroup's groups ascend and the smallest uncounted group is always being scanned.  Measured: no difference in time (r06_c44_pair_nctr.txt).
    static const uint32_t pairNctr = getenv("CXG_PAIR_NCTR") ? static_cast<uint32_t>(atoi(getenv("CXG_PAIR_NCTR"))) : 1u;   // (A/B only: 2, 4 or 8 counters)
    a.pair_nctr = (a.static_groups && (pairNctr == 2u || pairNctr == 4u || pairNctr == 8u)) ? pairNctr : 1u;
    le = cxgdev::launch_scan_teddy_pair(a, static_cast<uint32_t>(a.ngroups < static_cast<uint64_t>(cus) ? a.ngroups : static_cast<uint64_t>(cus)), stream);
  }
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
  const uint32_t kernelId = static_cast<uint32_t>(gen == 12 ? CXG_K_TEDDY_PAIR : gen == 11 ? CXG_K_DELIM_WAVE : trioKernel ? (persKernel ? CXG_K_TRIO_PERS : CXG_K_TRIO_WAVE) : litKernel ? CXG_K_LITERAL_PERS : persKernel ? CXG_K_FIELDS_PERS : fieldsKernel ? CXG_K_FIELDS_WAVE : (gen == 10 && fsmDirectRan) ? (fsmDirectTables ? CXG_K_FSM_DIRECT : CXG_K_FSM_LEAN) : gen >= 6 ? gen
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
    if (gen == 10 && ((err >> 8) & 0x32u) != 0u && ((err >> 8) & ~0x72u) == 0u && fsmMode < 3) {   // transducer kernel: row / event buffers overflowed
      // (0x40 — a row without a start — beside an overflow bit is a consequence of the dropped rows, not a finding)
      // 0x20 alone: the wave's row list -> mode 1 (2 tiles per wave); a sub-chunk's own buffers (0x02 rows, 0x10 events), or
      // mode 1 was not enough -> mode 2 (1 tile, 1 024 rows, 16 rows / 32 events per 32 bytes) -> mode 3 (2 048 rows; two workgroups per CU)
      fsmMode = ((err >> 8) == 0x20u && fsmMode == 0) ? 1 : (fsmMode < 2 ? 2 : 3);
      if (verbose) fprintf(stderr, "[cxg] transducer kernel: match-dense input (reason bits 0x%x), rerunning in mode %d\n", err >> 8, fsmMode);
      {                                                             // remembered per program; only grows
        uint8_t old = p->fsmMode[submatch ? 1 : 0].load(std::memory_order_relaxed);
        while (old < fsmMode && !p->fsmMode[submatch ? 1 : 0].compare_exchange_weak(old, static_cast<uint8_t>(fsmMode), std::memory_order_relaxed)) {}
      }
      relaunches++;
      continue;
    }
    if (gen == 12) {                                                // the pair kernel's budgets: the wave kernel (three-byte fingerprint, its own dense mode) for this program from now on
      if (verbose) fprintf(stderr, "[cxg] pair kernel raised the fallback flag (reason bits 0x%x): rerunning on scan_teddy_wave.hip\n", err >> 8);
      p->noPair[submatch ? 1 : 0].store(1, std::memory_order_relaxed);
      lastReason = err >> 8;
      relaunches++; gen = 7; continue;
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


}  // namespace cxgapi
