/* shim_harness.c — the cgo shim of INTEGRATION.md section 1, statement for statement, in C.
 *
 * There is no Go toolchain in the build image, so this program is the compilable stand-in for
 * integration/go/meta/findall_hip.go: the same three constructors (cxg_program_from_nfa / _from_literals / _from_charclass) fed
 * from freshly malloc'ed arrays (what flattenNFA does with C.malloc), the same cxg_find_all retry loop on
 * CXG_E_CAPACITY, the same FindAllSubmatch call.  tests/test_boundary.py runs it on the GPU and compares its
 * output with the oracle.
 *
 *   gcc -std=c99 -Iinclude examples/shim_harness.c -Lcoregex_amd -lcoregex_hip_rocm -o shim_harness
 *   ./shim_harness nfa       '\d+\.\d+\.\d+\.\d+'        access.log        # UseDigitPrefilter / UseDFA / UseBoth
 *   ./shim_harness submatch  '(\w+)@(\w+)\.(\w+)'         mail.log          # FindAllSubmatch hook (config 5)
 *   ./shim_harness literals  'error,warning,fatal'        app.log           # UseTeddy: prefilter.Teddy patterns
 *   ./shim_harness charclass '0-9,A-Z,_-_,a-z'            words.txt         # UseCharClassSearcher membership
 *
 * In `nfa` / `submatch` mode cxg_compile plays meta.Compile (it yields e.nfa, e.strategy, the two flags); everything
 * after that goes through the constructor the Go shim would call, never through the compiled program itself.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "coregex_hip.h"

static uint8_t* read_file(const char* path, uint64_t* len) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* b = (uint8_t*)malloc(n > 0 ? (size_t)n : 1);
  if (fread(b, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read\n"); exit(2); }
  fclose(f);
  *len = (uint64_t)n;
  return b;
}

/* The functions below mirror integration/go/meta/findall_hip.go one to one (same names in snake case, same order of
 * library calls): hip_init, flatten_nfa, nfa_program, literals_program, build_hip_program, find_all_hip, count_hip,
 * find_all_submatch_hip. */

typedef struct { cxg_program* p; cxg_program* sub; int ng; } hip_program;   /* type hipProgram */

static void hip_init(void) {                       /* func init(): refuse a library with another ABI */
  if (cxg_abi_version() != CXG_ABI_VERSION || cxg_timing_size() != sizeof(cxg_timing)) {
    fprintf(stderr, "coregex_hip: library ABI differs from the header this file was compiled against\n");
    exit(5);
  }
}

/* flattenNFA: nfa.State fields copied one by one into fresh memory the library may not keep. */
static cxg_nfa flatten_nfa(const cxg_nfa* e_nfa) {
  cxg_nfa n = *e_nfa;
  cxg_nfa_state* st = (cxg_nfa_state*)malloc(sizeof(cxg_nfa_state) * (n.n_states ? n.n_states : 1));
  cxg_nfa_trans* tr = (cxg_nfa_trans*)malloc(sizeof(cxg_nfa_trans) * (n.n_trans ? n.n_trans : 1));
  memcpy(st, e_nfa->states, sizeof(cxg_nfa_state) * n.n_states);
  if (n.n_trans) memcpy(tr, e_nfa->trans, sizeof(cxg_nfa_trans) * n.n_trans);
  n.states = st;
  n.trans = tr;
  return n;
}

/* nfaProgram(strategy, captures): capture_count 1 for the FindAllIndex / Count program, the NFA's own for FindAllSubmatch */
static cxg_program* nfa_program(const cxg_nfa* e_nfa, int strategy, uint32_t flags, int captures) {
  cxg_nfa n = flatten_nfa(e_nfa);
  if (!captures) n.capture_count = 1;
  cxg_program* prog = NULL;
  const int rc = cxg_program_from_nfa(&n, strategy, flags, &prog);
  free((void*)n.states);                            /* the library keeps nothing past return */
  free((void*)n.trans);
  if (rc != CXG_OK) { fprintf(stderr, "from_nfa: %d %s\n", rc, cxg_last_error()); return NULL; }
  return prog;
}

/* literalsProgram: prefilter.Teddy / FatTeddy patterns in pattern-ID order, all in C memory */
static cxg_program* literals_program(char* list, uint32_t* n_out) {
  const uint8_t* ptrs[64];
  uint32_t lens[64];
  uint32_t n = 0;
  for (char* tok = strtok(list, ","); tok && n < 64; tok = strtok(NULL, ",")) {
    uint8_t* c = (uint8_t*)malloc(strlen(tok) + 1);
    memcpy(c, tok, strlen(tok) + 1);
    ptrs[n] = c; lens[n] = (uint32_t)strlen(tok); n++;
  }
  cxg_program* prog = NULL;
  const int rc = cxg_program_from_literals(ptrs, lens, n, &prog);
  for (uint32_t i = 0; i < n; i++) free((void*)ptrs[i]);
  if (rc != CXG_OK) { fprintf(stderr, "from_literals: %d %s\n", rc, cxg_last_error()); exit(1); }
  *n_out = n;
  return prog;
}

/* buildHipProgram: the switch over e.strategy.  `mode` stands for what the compiled Engine holds: an NFA-carrying strategy
 * (cxg_compile plays meta.Compile and yields e.nfa, e.strategy, the flags), a Teddy prefilter, or a CharClassSearcher. */
static hip_program build_hip_program(const char* mode, char* spec) {
  hip_program h = {NULL, NULL, 1};
  if (cxg_device_count() <= 0) { fprintf(stderr, "no gfx950 device: e.hip stays nil\n"); exit(6); }
  if (!strcmp(mode, "literals")) {                  /* case UseTeddy */
    uint32_t n = 0;
    h.p = literals_program(spec, &n);
    printf("# strategy %s, %u literals\n", cxg_strategy_name(cxg_program_strategy(h.p)), n);
  } else if (!strcmp(mode, "charclass")) {          /* case UseCharClassSearcher: Membership(), MinMatch() */
    uint8_t m[256];
    memset(m, 0, sizeof m);
    for (char* tok = strtok(spec, ","); tok; tok = strtok(NULL, ",")) {
      if (strlen(tok) != 3 || tok[1] != '-') { fprintf(stderr, "range must be X-Y: %s\n", tok); exit(2); }
      for (int b = (uint8_t)tok[0]; b <= (uint8_t)tok[2]; b++) m[b] = 1;
    }
    const int rc = cxg_program_from_charclass(m, 1, &h.p);
    if (rc != CXG_OK) { fprintf(stderr, "from_charclass: %d %s\n", rc, cxg_last_error()); exit(1); }
    printf("# strategy %s\n", cxg_strategy_name(cxg_program_strategy(h.p)));
  } else {                                          /* case UseDigitPrefilter, UseDFA, UseBoth, UseNFA, UseBoundedBacktracker, UseTeddy behind (?m)^ */
    cxg_program* engine = NULL;                     /* stands for the compiled *meta.Engine */
    int rc = cxg_compile(spec, strlen(spec), &engine);
    if (rc != CXG_OK) { fprintf(stderr, "compile: %d %s\n", rc, cxg_last_error()); exit(1); }
    const int strategy = cxg_program_strategy(engine);
    switch (strategy) {
      case CXG_USE_DIGIT_PREFILTER: case CXG_USE_DFA: case CXG_USE_BOTH: case CXG_USE_NFA: case CXG_USE_BOUNDED_BACKTRACKER: case CXG_USE_TEDDY: break;
      default:
        fprintf(stderr, "strategy %s carries no NFA for the device (use literals / charclass mode)\n", cxg_strategy_name(strategy));
        exit(3);
    }
    cxg_nfa e_nfa;
    rc = cxg_program_nfa(engine, &e_nfa);
    if (rc != CXG_OK) { fprintf(stderr, "nfa: %d %s\n", rc, cxg_last_error()); exit(1); }
    const uint32_t flags = cxg_program_flags(engine);                /* e.digitRunSkipSafe, e.reverseDFA != nil, e.prefilter != nil */
    h.ng = (int)e_nfa.capture_count;
    h.p = nfa_program(&e_nfa, strategy, flags, 0);
    if (h.p && cxg_program_supported(h.p) != 1) { cxg_program_destroy(h.p); h.p = NULL; }
    if (e_nfa.capture_count > 1) {                                   /* FindAllSubmatch: the same NFA with its real capture count */
      h.sub = nfa_program(&e_nfa, strategy, flags, 1);
      if (h.sub && cxg_program_submatch_supported(h.sub) != 1) { cxg_program_destroy(h.sub); h.sub = NULL; }
    }
    cxg_program_destroy(engine);                                     /* neither program depends on the engine's memory */
    printf("# strategy %s, %d group(s)\n", cxg_strategy_name(strategy), h.ng);
    return h;
  }
  if (h.p && cxg_program_supported(h.p) != 1) { cxg_program_destroy(h.p); h.p = NULL; }
  return h;
}

/* findAllHip: results := make([][2]int, 0, len(haystack)/100+1); grow and retry on CXG_E_CAPACITY; anything else: the CPU loop */
static int find_all_hip(const hip_program* h, const uint8_t* hay, uint64_t len, int64_t limit, int64_t** rows_io, uint64_t* got, int* retries) {
  if (!h->p) return 0;
  uint64_t cap = len / 100 + 1;
  int64_t* rows = (int64_t*)malloc(cap * 2 * sizeof(int64_t));
  for (;;) {
    const int rc = cxg_find_all(h->p, hay, len, limit, rows, cap, got);
    if (rc == CXG_OK) { *rows_io = rows; return 1; }
    if (rc == CXG_E_CAPACITY) { cap = *got; rows = (int64_t*)realloc(rows, cap * 2 * sizeof(int64_t)); (*retries)++; continue; }
    fprintf(stderr, "find_all: %d %s\n", rc, cxg_last_error());
    free(rows);
    return 0;
  }
}

/* countHip */
static int count_hip(const hip_program* h, const uint8_t* hay, uint64_t len, int64_t limit, uint64_t* got) {
  return h->p && cxg_count(h->p, hay, len, limit, got) == CXG_OK;
}

/* findAllSubmatchHip: rows of 2 * groups int64, -1 for a group that did not take part */
static int find_all_submatch_hip(const hip_program* h, const uint8_t* hay, uint64_t len, int64_t limit, int64_t** rows_io, uint64_t* got, int* retries) {
  if (!h->sub) return 0;
  const uint64_t width = 2u * (uint64_t)h->ng;
  uint64_t cap = len / 100 + 1;
  int64_t* rows = (int64_t*)malloc(cap * width * sizeof(int64_t));
  for (;;) {
    const int rc = cxg_find_all_submatch(h->sub, hay, len, limit, rows, cap, got);
    if (rc == CXG_OK) { *rows_io = rows; return 1; }
    if (rc == CXG_E_CAPACITY) { cap = *got; rows = (int64_t*)realloc(rows, cap * width * sizeof(int64_t)); (*retries)++; continue; }
    fprintf(stderr, "find_all_submatch: %d %s\n", rc, cxg_last_error());
    free(rows);
    return 0;
  }
}

/* Thread hygiene (round 6; findall_hip.go's worker pool): the library's scratch is per OS thread.  `threads N SPEC FILE`: N short-lived
 * threads, one after another in groups of 8, each runs find_all_hip once over FILE and exits; the device's free memory is printed before,
 * after the threads are gone, and after the main thread's own cxg_thread_release — thread_local destructors must have handed everything
 * back (what the Go runtime's thread churn would otherwise leave behind). */
#include <pthread.h>
typedef struct { const hip_program* h; const uint8_t* hay; uint64_t len; uint64_t got; int ok; } thread_job;
static void* thread_main(void* arg) {
  thread_job* j = (thread_job*)arg;
  int64_t* rows = NULL;
  int retries = 0;
  j->ok = find_all_hip(j->h, j->hay, j->len, -1, &rows, &j->got, &retries);
  free(rows);
  return NULL;                                                 /* no cxg_thread_release: the thread's exit must do it */
}
static int threads_mode(int nthreads, char* spec, const char* file) {
  hip_init();
  uint64_t len = 0;
  uint8_t* hay = read_file(file, &len);
  hip_program h = build_hip_program("nfa", spec);
  if (!h.p) { fprintf(stderr, "device path refuses the program: %s\n", cxg_last_error()); return 4; }
  uint64_t want = 0, f0 = 0, f1 = 0, f2 = 0, tot = 0;
  { int64_t* rows = NULL; int retries = 0; if (!find_all_hip(&h, hay, len, -1, &rows, &want, &retries)) return 1; free(rows); }   /* main thread: context, program image, its own scratch */
  cxg_device_mem_info(0, &f0, &tot);
  for (int base = 0; base < nthreads; base += 8) {
    pthread_t th[8];
    thread_job job[8];
    const int n = nthreads - base < 8 ? nthreads - base : 8;
    for (int i = 0; i < n; i++) { job[i].h = &h; job[i].hay = hay; job[i].len = len; job[i].got = 0; job[i].ok = 0; pthread_create(&th[i], NULL, thread_main, &job[i]); }
    for (int i = 0; i < n; i++) { pthread_join(th[i], NULL); if (!job[i].ok || job[i].got != want) { fprintf(stderr, "thread %d: %d rows, want %llu\n", base + i, (int)job[i].got, (unsigned long long)want); return 1; } }
  }
  cxg_device_mem_info(0, &f1, &tot);
  cxg_thread_release();
  cxg_device_mem_info(0, &f2, &tot);
  printf("# %d threads x %llu rows; device free bytes before %llu, after the threads %llu, after cxg_thread_release %llu\n", nthreads, (unsigned long long)want,
         (unsigned long long)f0, (unsigned long long)f1, (unsigned long long)f2);
  printf("leaked_by_threads %lld\n", (long long)f0 - (long long)f1);
  cxg_program_destroy(h.p);
  free(hay);
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 5 && !strcmp(argv[1], "threads")) return threads_mode(atoi(argv[2]), argv[3], argv[4]);
  if (argc != 4) { fprintf(stderr, "usage: %s nfa|submatch|literals|charclass SPEC FILE | threads N SPEC FILE\n", argv[0]); return 2; }
  const char* mode = argv[1];
  if (strcmp(mode, "nfa") && strcmp(mode, "submatch") && strcmp(mode, "literals") && strcmp(mode, "charclass")) { fprintf(stderr, "unknown mode %s\n", mode); return 2; }
  hip_init();
  uint64_t len = 0;
  uint8_t* hay = read_file(argv[3], &len);
  hip_program h = build_hip_program(mode, argv[2]);
  const int sub = !strcmp(mode, "submatch");
  if (sub ? h.sub == NULL : h.p == NULL) {
    fprintf(stderr, "device path refuses the program: %s\n", cxg_last_error());   /* e.hip stays nil */
    return 4;
  }
  const uint64_t width = sub ? 2u * (uint64_t)h.ng : 2u;
  int64_t* rows = NULL;
  uint64_t got = 0;
  int retries = 0;
  const int ok = sub ? find_all_submatch_hip(&h, hay, len, -1, &rows, &got, &retries) : find_all_hip(&h, hay, len, -1, &rows, &got, &retries);
  if (!ok) return 1;                                /* degrade: the Go caller runs its CPU loop */
  printf("# %llu rows, %d capacity retries\n", (unsigned long long)got, retries);
  for (uint64_t i = 0; i < got; i++) {
    for (uint64_t k = 0; k < width; k++) printf(k ? " %lld" : "%lld", (long long)rows[i * width + k]);
    printf("\n");
  }
  uint64_t cnt = 0;
  if (!sub && (!count_hip(&h, hay, len, -1, &cnt) || cnt != got)) { fprintf(stderr, "count mismatch\n"); return 1; }
  free(rows);
  free(hay);
  if (h.p) cxg_program_destroy(h.p);
  if (h.sub) cxg_program_destroy(h.sub);
  cxg_thread_release();
  return 0;
}
