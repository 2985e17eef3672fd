// capi.hip — the extern "C" surface of libcoregex_hip.so (include/coregex_hip.h).
// Host glue only: program construction (host/), device copies, per-thread stream + scratch
// (the SearchState analogue, meta/search_state.go:23-62), launches (device/).  There is no CPU
// search path in this library: without a gfx950 device every search entry returns CXG_E_NO_GPU.
// (Round 6: the pieces behind these functions live in capi_state / capi_captures / capi_ladder / capi_nullable / capi_host .hip, shared
// declarations in capi_internal.hpp.)
#include "capi_internal.hpp"

using namespace cxgapi;

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
    case CXG_K_TEDDY_PAIR: return "k_scan_teddy_pair";
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
  {
    OrderGate orderGate(g_path[b->device], nullptr);               // (a fill beside another thread's persistent scan would be a foreign kernel to it)
    HIP_TRY(launchFillSynth(b->d, npages, config, seed, first_page));
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
