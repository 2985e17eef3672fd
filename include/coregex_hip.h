/* coregex_hip.h — C ABI of libcoregex_hip.so: the MI355X (gfx950) bulk FindAll path.
 *
 * Drop-in boundary.  The reference has no FFI today; the seam these entry
 * points replace is one call per *haystack*:
 *
 *   cxg_find_all           <- (*meta.Engine).FindAllIndicesStreaming   meta/findall.go:155
 *                              (reached from Regex.FindAll/FindAllIndex/AppendAllIndex, regex.go:395,702,752)
 *   cxg_count              <- (*meta.Engine).Count                      meta/findall.go:297
 *   cxg_find_all_submatch  <- (*meta.Engine).FindAllSubmatch            meta/findall.go:390
 *                              (flattened as Regex.FindAllSubmatchIndex does, regex.go:1423-1450)
 *   cxg_program_from_nfa   <- what a cgo shim builds once per compiled *meta.Engine from
 *                              e.nfa (nfa/nfa.go:23-154 State), e.strategy (meta/strategy.go:19-230),
 *                              e.digitRunSkipSafe (meta/compile.go:176)
 *   cxg_program_from_literals <- prefilter.Teddy patterns (prefilter/teddy.go:110-130) for UseTeddy
 *   cxg_program_from_charclass <- nfa.CharClassSearcher.membership (nfa/charclass_searcher.go:21-27)
 *   cxg_compile            <- meta.Compile (meta/compile.go:40): host stand-in used where no Go
 *                              toolchain exists (this image); same pattern -> strategy -> tables
 *                              pipeline, C++ (coregex_amd/csrc/host/).
 *
 * Conventions: return 0 on success, a negative CXG_E_* otherwise.  Offsets are absolute int64
 * indices into the haystack, identical to Go `int`.  The caller owns `hay` and the output
 * arrays; nothing is retained past return (cgo pointer rule).  A cxg_program is immutable and
 * may be shared by threads; every call uses its own stream and scratch (the SearchState
 * analogue, meta/search_state.go:23-62).  No torch types appear in any signature.
 */
#ifndef COREGEX_HIP_H_
#define COREGEX_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CXG_OK 0
#define CXG_E_INVALID (-1)      /* bad argument / malformed program description */
#define CXG_E_UNSUPPORTED (-2)  /* pattern or strategy outside the accelerated subset: caller keeps its CPU loop */
#define CXG_E_CAPACITY (-3)     /* output array too small: *n_out holds the required number of rows */
#define CXG_E_DEVICE (-4)       /* HIP runtime failure (cxg_last_error has the text) */
#define CXG_E_NO_GPU (-5)       /* no gfx950 device visible: there is no CPU fallback in this library */
#define CXG_E_SYNTAX (-6)       /* cxg_compile: pattern does not parse */
#define CXG_E_INTERNAL (-7)     /* device-side invariant violated (watchdog, scratch overflow) */
#define CXG_E_INPUT (-8)        /* THIS haystack cannot be answered on the device; the caller keeps its CPU loop for this
                                   call, the program stays usable.  Two cases: (a) a UseBoth program WITH ASSERTIONS met a match
                                   longer than 100 bytes, or any UseBoth program met more than 64 of them in one haystack (the
                                   reference restarts its PikeVM inside such a match, meta/find_indices.go:425-431; without
                                   assertions the device path restarts its search at the same place, round 3);
                                   (b) a program WITHOUT a transducer image (cxg_program_fsm_image: > 224 stack states, or
                                   a table beyond the LDS budget) met > 128 KiB without a synchronising byte.  Programs
                                   with the image have no such limit (64 MiB of "1.1.1.1..." is answered, tests). */
#define CXG_E_THREAD (-9)       /* cxg_wait on another thread than the one that made the asynchronous call: the handle is untouched, wait there */

/* meta.Strategy values (meta/strategy.go:19-230), same numbering. */
enum cxg_strategy {
  CXG_USE_NFA = 0, CXG_USE_DFA = 1, CXG_USE_BOTH = 2, CXG_USE_REVERSE_ANCHORED = 3,
  CXG_USE_REVERSE_SUFFIX = 4, CXG_USE_ONEPASS = 5, CXG_USE_REVERSE_INNER = 6,
  CXG_USE_BOUNDED_BACKTRACKER = 7, CXG_USE_TEDDY = 8, CXG_USE_REVERSE_SUFFIX_SET = 9,
  CXG_USE_CHARCLASS_SEARCHER = 10, CXG_USE_COMPOSITE_SEARCHER = 11, CXG_USE_BRANCH_DISPATCH = 12,
  CXG_USE_DIGIT_PREFILTER = 13, CXG_USE_AHO_CORASICK = 14, CXG_USE_ANCHORED_LITERAL = 15,
  CXG_USE_MULTILINE_REVERSE_SUFFIX = 16
};

/* nfa.StateKind (nfa/nfa.go:23-60), same numbering. */
enum cxg_nfa_kind {
  CXG_NFA_MATCH = 0, CXG_NFA_BYTE_RANGE = 1, CXG_NFA_SPARSE = 2, CXG_NFA_SPLIT = 3,
  CXG_NFA_EPSILON = 4, CXG_NFA_CAPTURE = 5, CXG_NFA_FAIL = 6, CXG_NFA_LOOK = 7,
  /* nfa.StateRuneAny / StateRuneAnyNotNL (nfa/nfa.go:53-59): states of the PikeVM's rune NFA only, never of Engine.nfa.  Known to the
     library so that a binding which meets one gets CXG_E_UNSUPPORTED (the caller keeps its CPU loop), not CXG_E_INVALID. */
  CXG_NFA_RUNE_ANY = 8, CXG_NFA_RUNE_ANY_NOT_NL = 9
};

#define CXG_NFA_INVALID 0xFFFFFFFFu

typedef struct cxg_nfa_trans {  /* nfa.Transition (nfa/nfa.go:138-142) */
  uint8_t lo, hi;
  uint16_t _pad;
  uint32_t next;
} cxg_nfa_trans;

typedef struct cxg_nfa_state {  /* nfa.State (nfa/nfa.go:119-154), flattened */
  uint8_t kind;        /* cxg_nfa_kind */
  uint8_t lo, hi;      /* BYTE_RANGE; LOOK: lo = nfa.Look (nfa/nfa.go:92-117: 0 StartText, 1 EndText, 2 StartLine,
                          3 EndLine, 4 WordBoundary, 5 NoWordBoundary) */
  uint8_t cap_start;   /* CAPTURE: 1 = opening */
  uint32_t next;       /* BYTE_RANGE / EPSILON / CAPTURE / LOOK */
  uint32_t left, right;/* SPLIT (left is explored first) */
  uint32_t cap_index;  /* CAPTURE */
  uint32_t trans_off;  /* SPARSE: first entry in the transition array */
  uint32_t trans_len;  /* SPARSE */
} cxg_nfa_state;

typedef struct cxg_nfa {
  const cxg_nfa_state* states;
  uint32_t n_states;
  const cxg_nfa_trans* trans;
  uint32_t n_trans;
  uint32_t start_anchored, start_unanchored;  /* NFA.StartAnchored / StartUnanchored */
  uint32_t capture_count;                      /* NFA.CaptureCount(), includes group 0 */
} cxg_nfa;

#define CXG_FLAG_DIGIT_RUN_SKIP_SAFE 1u  /* Engine.digitRunSkipSafe (meta/compile.go:176) */
#define CXG_FLAG_HAS_REVERSE_DFA 2u      /* Engine.reverseDFA != nil (meta/compile.go:184-205) */
#define CXG_FLAG_HAS_PREFILTER 4u        /* Engine.prefilter != nil (meta/compile.go:466-478; prefilter/prefilter.go:261-297: one prefix literal, or 2+ of >= 3 bytes).
                                            UseBoth: with a prefilter the reference runs its PikeVM from the prefilter's position (plain leftmost-first,
                                            find_indices.go:411-429); without one it restarts the PikeVM at end - 100 inside a longer match (:432-441). */

typedef struct cxg_program cxg_program;  /* opaque: strategy + tables, host and device copies */
typedef struct cxg_buffer cxg_buffer;    /* opaque: device-resident haystack */

typedef struct cxg_timing {  /* filled by the *_device entry points when non-NULL */
  float kernel_ms;           /* HIP-event time of the scan kernel(s) on the call's stream */
  float total_ms;            /* scan + status reset + count read-back, same stream */
  uint32_t n_launches;
  uint32_t grid, block;
  uint64_t tiles;
  uint32_t kernel;           /* cxg_kernel id of the LAST scan launch (after fallbacks): see cxg_kernel_name */
  uint32_t fallback_reason;  /* reason bits of the last fallback a wave kernel raised in this call, 0: none */
  uint32_t n_ladder;         /* span launches of this call (a fallback adds a rung: wave kernel -> denser mode -> transducer -> table kernel) */
  uint8_t ladder[12];        /* cxg_kernel id of each of them, in order (the first 12) */
} cxg_timing;

/* Kernel families (cxg_timing.kernel). */
enum cxg_kernel {
  CXG_K_NONE = 0, CXG_K_DFA_TABLE = 1, CXG_K_DIGIT_FLAT = 2, CXG_K_CHAIN_WAVE = 6, CXG_K_TEDDY_WAVE = 7,
  CXG_K_CHARCLASS_WAVE = 8, CXG_K_PREFIX_WAVE = 9, CXG_K_FSM = 10, CXG_K_TEDDY_TABLE = 11, CXG_K_CHARCLASS_TABLE = 12,
  CXG_K_FIELDS_WAVE = 13,  /* scan_fields_wave.hip: fields programs such as `\d+\.\d+\.\d+\.\d+` (round 3) */
  CXG_K_TRIO_WAVE = 14,    /* scan_fields_wave.hip k_scan_trio_wave: run a run b run programs such as `(\w+)@(\w+)\.(\w+)` (round 3) */
  CXG_K_FIELDS_PERS = 15,  /* scan_fields_wave.hip k_scan_fields_pers: the fields mathematics on a persistent grid, ordering deferred by a round (round 4) */
  CXG_K_DELIM_WAVE = 16,   /* scan_delim_wave.hip: `O [^E]+ E` programs such as `\[[^\]]+\]` (round 4) */
  CXG_K_LITERAL_PERS = 17, /* k_scan_fields_pers in its literal mode: border-free literals over <= 4 distinct bytes such as `error` (round 5) */
  CXG_K_TRIO_PERS = 18,    /* k_scan_fields_pers in its TRIO mode: k_scan_trio_wave's programs (spans or capture rows) on the persistent grid (round 5) */
  CXG_K_FSM_DIRECT = 19,   /* scan_fsm.hip k_scan_fsml<direct>: the transducer with byte-indexed rows — one table read per byte, no class lookup (round 6) */
  CXG_K_FSM_LEAN = 20,     /* scan_fsm.hip k_scan_fsml: shallow machines on input whose entry states collapse, without k_scan_fsm's machinery for those that do not (round 6) */
  CXG_K_TEDDY_PAIR = 21    /* scan_teddy_pair.hip: literal sets with one fingerprint lookup per byte PAIR, persistent grid with claimed groups (round 6) */
};
const char* cxg_kernel_name(int kernel);

const char* cxg_last_error(void);          /* thread-local text of the last failure */
const char* cxg_version(void);
/* ABI of this header: bumped whenever a struct a caller allocates (cxg_timing, cxg_path_state_t, cxg_nfa*) changes size or
 * layout.  A binding compiled against another header must refuse to run: `cxg_abi_version() != CXG_ABI_VERSION` or
 * `cxg_timing_size() != sizeof(cxg_timing)`.  History: 1 = rounds 1-3; 2 = round 4 (cxg_timing grew by n_ladder + ladder[12]:
 * 16 bytes, every *_device entry point writes the whole struct); 3 = round 5 (cxg_path_state_t, cxg_pending). */
#define CXG_ABI_VERSION 3
int cxg_abi_version(void);
size_t cxg_timing_size(void);

/* Launch-mode state of a device (process-wide).  The fastest launch modes assume that the device dispatches workgroups in index
 * order (static groups, delimiter kernel) or holds a whole grid resident (persistent fields kernel); a spin watchdog catches the
 * cases where another tenant of the GPU breaks that.  A hit demotes the mode for `*_penalty` further calls (8, doubling to 1024
 * on repeated hits, reset by a clean call), then it is tried again; the call that was hit reruns one mode down and still
 * returns the reference's rows.  `*_hits` count watchdog hits since the process started. */
typedef struct cxg_path_state_t {
  uint32_t static_penalty, static_hits;
  uint32_t persistent_penalty, persistent_hits;
  uint32_t delim_penalty, delim_hits;
  uint32_t order_waiters;          /* threads waiting for the device's order-dependent launch slot (such launches take turns: two of them side by side can deadlock) */
  uint32_t reserved;
} cxg_path_state_t;
int cxg_path_state(int device, cxg_path_state_t* out);
/* Forget every demotion of `device` (penalties 0, terms back to 8): for a host that knows the co-tenant that caused them is gone. */
int cxg_path_reset(int device);
/* Diagnostics / tests: act as if the spin watchdog of a mode had fired on `device` (mode 0 static groups, 1 persistent grid,
 * 2 delimiter kernel): the mode is demoted for its current term, exactly as a real hit would. */
int cxg_debug_demote(int device, int mode);
int cxg_device_count(void);                /* gfx950 devices visible; 0 => every search returns CXG_E_NO_GPU */
int cxg_device_mem_info(int device, uint64_t* free_bytes, uint64_t* total_bytes);   /* hipMemGetInfo of that device: a host sizes its shards with it, the
                                                                                    thread-hygiene test watches it (round 6) */
int cxg_set_device(int device);            /* per-thread device for subsequent calls (default 0) */
/* Frees the calling thread's stream, events, pinned buffers and HBM staging (they are per OS thread and are also freed
 * when the thread exits).  A host whose callers hop across threads (cgo) may call it when a thread goes idle. */
void cxg_thread_release(void);

/* ---- program construction ------------------------------------------------------------ */
int cxg_compile(const char* pattern, size_t len, cxg_program** out);
int cxg_program_from_nfa(const cxg_nfa* nfa, int strategy, uint32_t flags, cxg_program** out);
int cxg_program_from_literals(const uint8_t* const* lits, const uint32_t* lens, uint32_t n, cxg_program** out);
int cxg_program_from_charclass(const uint8_t membership[256], uint32_t min_match, cxg_program** out);
void cxg_program_destroy(cxg_program* p);

int cxg_program_strategy(const cxg_program* p);         /* cxg_strategy */
uint32_t cxg_program_flags(const cxg_program* p);       /* CXG_FLAG_* the program was built with (cxg_compile: what the
                                                           front-end derived, i.e. Engine.digitRunSkipSafe / reverseDFA != nil) */
const char* cxg_strategy_name(int strategy);
int cxg_program_num_groups(const cxg_program* p);       /* NumSubexp()+1 */
int cxg_program_nfa_states(const cxg_program* p);       /* -1 if built without an NFA */
int cxg_program_dfa_states(const cxg_program* p);       /* eager forward DFA states incl. dead */
int cxg_program_supported(const cxg_program* p);        /* 1 if the device path accepts it */
/* Nullable pattern (`a*`, `x?y*`: matches the empty string; meta/findall.go:251-275 is its FindAll rule)?  0 no; 1 the device
   program is the pattern's non-empty variant and the empty matches are merged behind the scan; 2 every match is empty (`a*?`). */
/* Offset captures: every capture boundary of the pattern lies a fixed number of bytes behind the match's start or in front of its
   end (`user=(\S+)`, `"([^"]*)"`); FindAllSubmatch is then FindAll + one expansion kernel.  Returns the number of slots (2 x groups)
   and, per slot, src (0 = start, 1 = end) and the delta added to it; 0 when the program has no such description. */
int cxg_program_offset_captures(const cxg_program* p, int* src, int* delta, int max_slots);
int cxg_program_nullable(const cxg_program* p);
/* `O [^E]+ E` / `O [^E]* E` program (`\[[^\]]+\]`, `<[^>]+>`: served by scan_delim_wave.hip in front of the transducer)?  1 and the two
   bytes + whether the class must be taken at least once; else 0. */
int cxg_program_delimiters(const cxg_program* p, int* open_byte, int* close_byte, int* plus);
/* Device image of the program (what every kernel stages into LDS); for tests and the emulator. */
int cxg_program_blob(const cxg_program* p, const void** data, size_t* len);
/* Diagnostics: the FindAll transducer image of the general-DFA kernel (coregex_amd/csrc/device/fsm.hpp): of the
 * FindAllIndex program (submatch == 0) or of the span program of FindAllSubmatchIndex (submatch != 0).
 * CXG_E_UNSUPPORTED when the program has none (served by other kernels alone, or outside the table budget). */
int cxg_program_fsm_image(const cxg_program* p, int submatch, const void** data, size_t* len);
/* Images used by the FindAllSubmatch path: bidirectional-DFA span program + one-pass capture table. */
int cxg_program_submatch_blobs(const cxg_program* p, const void** span_blob, size_t* span_len,
                               const void** cap_blob, size_t* cap_len);
/* Diagnostics: the 40-byte ChainCaps record (device/walk.hpp) of a FindAllSubmatch program whose capture slots are
 * written by the chain kernel itself (slot = match start / match end / end of one of two runs, plus a constant).
 * Returns 1 and fills out[40] when the program has one, 0 otherwise (captures then come from the one-pass table). */
int cxg_program_chain_captures(const cxg_program* p, uint8_t out[40]);
/* Diagnostics: the field bounds of a bounded-repetition program served by the chain kernel (`\d{1,3}\.\d{1,3}`...): same
 * 40-byte record with on == 2, nruns = fields but the last, src[x] = min and src[8 + x] = max (0: unbounded) of field x.
 * Returns 1 when the program has one. */
int cxg_program_chain_bounds(const cxg_program* p, uint8_t out[40]);
/* 1 when cxg_find_all_submatch is served.  Independent of cxg_program_supported for an NFA with assertions: FindAllSubmatch
 * of a UseNFA / UseDFA / UseBoth / UseDigitPrefilter / UseBoundedBacktracker engine is the PikeVM over the whole haystack
 * (meta/findall.go:89-98), which the device reproduces (look-around transducer for the spans, backtracking pass with the
 * assertions for the slots) whatever that engine's lazy DFA does for FindAllIndex. */
int cxg_program_submatch_supported(const cxg_program* p);
/* Host-side copy of the NFA a program was compiled from (cxg_compile only); pointers live as long as p. */
int cxg_program_nfa(const cxg_program* p, cxg_nfa* out);

/* ---- search over a host haystack (what the cgo shim calls) --------------------------- */
int cxg_find_all(const cxg_program* p, const uint8_t* hay, uint64_t len, int64_t limit,
                 int64_t* spans /* [cap][2] */, uint64_t cap, uint64_t* n_out);
int cxg_count(const cxg_program* p, const uint8_t* hay, uint64_t len, int64_t limit, uint64_t* n_out);
/* (*Engine).Find (meta/find.go:29) and (*Engine).IsMatch (meta/ismatch.go:27) of a whole haystack (round 6): the first match in haystack
 * order — FindAll with n == 1, whose early stop lets workgroups that start after the first row has been counted leave at once.  *found /
 * *matched are 0 or 1; span = (start, end) of the match when found.  Same error conventions as cxg_find_all: anything but CXG_OK means
 * "run the CPU search". */
int cxg_find(const cxg_program* p, const uint8_t* hay, uint64_t len, int64_t span[2], int* found);
int cxg_is_match(const cxg_program* p, const uint8_t* hay, uint64_t len, int* matched);
int cxg_find_all_submatch(const cxg_program* p, const uint8_t* hay, uint64_t len, int64_t limit,
                          int64_t* slots /* [cap][2*groups], -1 unset */, uint64_t cap, uint64_t* n_out);

/* ---- device-resident corpus (benchmarks, multi-GPU shards) --------------------------- */
int cxg_buffer_alloc(uint64_t len, cxg_buffer** out);               /* on the calling thread's device */
void cxg_buffer_free(cxg_buffer* b);
int cxg_buffer_upload(cxg_buffer* b, uint64_t off, const uint8_t* src, uint64_t len);
int cxg_buffer_download(const cxg_buffer* b, uint64_t off, uint8_t* dst, uint64_t len);
uint64_t cxg_buffer_len(const cxg_buffer* b);
void* cxg_buffer_device_ptr(const cxg_buffer* b);
/* synthlog-v1 (DESIGN.md "Synthetic corpus"): page `first_page + i` of config `config` written at
 * byte offset i*4096.  Generated on the device; cxg_synth_page_host is the CPU twin used by tests. */
int cxg_buffer_fill_synth(cxg_buffer* b, uint32_t config, uint64_t seed, uint64_t first_page);
int cxg_synth_page_host(uint32_t config, uint64_t seed, uint64_t page, uint8_t out[4096]);

/* Raw device pointers (torch tensors' data_ptr(), a HIP stream handle or NULL for the call's own).
 * d_out holds rows of `row_width` int64 (2 for find_all, 2*groups for submatch); rows are absolute
 * offsets into d_hay plus `base`.  d_out may be NULL with cap 0 to count only. */
int cxg_find_all_device(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit,
                        void* d_out, uint64_t cap, uint64_t* n_out, void* stream, cxg_timing* timing);
int cxg_find_all_submatch_device(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base,
                                 int64_t limit, void* d_out, uint64_t cap, uint64_t* n_out, void* stream,
                                 cxg_timing* timing);

int cxg_find_device(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t span[2], int* found, void* stream);
int cxg_is_match_device(const cxg_program* p, const void* d_hay, uint64_t len, int* matched, void* stream);

/* Asynchronous form of cxg_find_all_device (round 5): the call returns with its span launch in flight on `stream` (NULL: the calling
 * thread's own); cxg_wait — on the SAME thread — completes it and returns what cxg_find_all_device would have returned (a launch that
 * needs another kernel is rerun synchronously inside cxg_wait; programs without an async-capable first launch run to completion inside
 * the async call).  Up to 16 calls per thread may be pending, on any streams: since round 6 the launch sections of ALL threads are
 * chained on the device (each section's stream waits for the completion event of the section in front of it), so a pending call
 * holds no lock — other threads' scans simply queue behind it — and pending calls on different streams cannot overtake each other's
 * scratch words.  cxg_wait on another thread returns CXG_E_THREAD and leaves the handle valid; every other return frees it.
 * d_hay, d_out and the program must stay valid until cxg_wait returns.  Mirrors nothing in the reference (its calls are synchronous);
 * it is what a batch host (bench.py, a shard scheduler) uses to pay launch + sync once per batch. */
typedef struct cxg_pending cxg_pending;
int cxg_find_all_device_async(const cxg_program* p, const void* d_hay, uint64_t len, int64_t base, int64_t limit,
                              void* d_out, uint64_t cap, void* stream, cxg_pending** out);
int cxg_wait(cxg_pending* pending, uint64_t* n_out, cxg_timing* timing);   /* frees the handle; timing->kernel_ms is 0 unless CXG_ASYNC_TIMING
                                                                               is set in the environment (a start event per pending launch) */

/* Compact rows for shard-sized, device-resident haystacks: rows of two uint32 — (start, end) relative to d_hay, no `base` —
 * 8 bytes per match instead of 16.  Device-only entry point beside the int64 ABI above (which the cgo binding keeps using): a
 * consumer on the device (a later kernel, a gather that rebases per shard) reads half the bytes, and the write-bound programs —
 * nfa.CharClassSearcher.FindAllIndices (nfa/charclass_searcher.go:158-211) emits 16 bytes per run of ~5.5 bytes of log text —
 * write half of them.  len must be below 4 GiB.  Served for the programs whose span kernel has the compact row epilogue
 * (char-class programs incl. `\S+`-style class runs; fields programs such as `\d+\.\d+\.\d+\.\d+`); CXG_E_UNSUPPORTED
 * otherwise, CXG_E_INPUT when the haystack needs a kernel without it (the caller uses cxg_find_all_device). */
int cxg_find_all_device_u32(const cxg_program* p, const void* d_hay, uint64_t len, int64_t limit, void* d_out_u32,
                            uint64_t cap, uint64_t* n_out, void* stream, cxg_timing* timing);

#ifdef __cplusplus
}
#endif
#endif /* COREGEX_HIP_H_ */
