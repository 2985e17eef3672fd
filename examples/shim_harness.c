/* shim_harness.c — the cgo shim of INTEGRATION.md section 1, statement for statement, in C.
 *
 * There is no Go toolchain in the build image, so this program is the compilable stand-in for
 * meta/findall_hip.go: the same three constructors (cxg_program_from_nfa / _from_literals / _from_charclass) fed
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

/* flattenNFA: nfa.State fields copied one-to-one into C memory. */
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

/* buildHipProgram, strategies that carry an NFA */
static cxg_program* build_from_engine(const char* pattern) {
  cxg_program* engine = NULL;                       /* stands for the compiled *meta.Engine */
  int rc = cxg_compile(pattern, strlen(pattern), &engine);
  if (rc != CXG_OK) { fprintf(stderr, "compile: %d %s\n", rc, cxg_last_error()); exit(1); }
  const int strategy = cxg_program_strategy(engine);
  cxg_program* prog = NULL;
  switch (strategy) {
    case CXG_USE_DIGIT_PREFILTER: case CXG_USE_DFA: case CXG_USE_BOTH:
    case CXG_USE_NFA:                                          /* word boundaries / multi-line anchors: nfa.StateLook, lo = nfa.Look */
    case CXG_USE_TEDDY: {                                      /* reached in this mode by (?m)^ literal alternations (lineAnchorWrapper) */
      cxg_nfa e_nfa;
      rc = cxg_program_nfa(engine, &e_nfa);
      if (rc != CXG_OK) { fprintf(stderr, "nfa: %d %s\n", rc, cxg_last_error()); exit(1); }
      cxg_nfa n = flatten_nfa(&e_nfa);
      const uint32_t flags = cxg_program_flags(engine);   /* e.digitRunSkipSafe, e.reverseDFA != nil */
      cxg_program_destroy(engine);                  /* the constructor must not depend on the engine's memory */
      engine = NULL;
      rc = cxg_program_from_nfa(&n, strategy, flags, &prog);
      free((void*)n.states);                        /* ... nor on the flattened arrays after it returns */
      free((void*)n.trans);
      if (rc != CXG_OK) { fprintf(stderr, "from_nfa: %d %s\n", rc, cxg_last_error()); exit(1); }
      break;
    }
    default:
      fprintf(stderr, "strategy %s carries no NFA for the device (use literals / charclass mode)\n", cxg_strategy_name(strategy));
      exit(3);
  }
  if (engine) cxg_program_destroy(engine);
  printf("# strategy %s, %d group(s)\n", cxg_strategy_name(strategy), cxg_program_num_groups(prog));
  return prog;
}

static cxg_program* build_from_literals(char* list) {
  const uint8_t* ptrs[64];
  uint32_t lens[64];
  uint32_t n = 0;
  for (char* tok = strtok(list, ","); tok && n < 64; tok = strtok(NULL, ",")) {
    uint8_t* c = (uint8_t*)malloc(strlen(tok) + 1);  /* pinned / C memory, as the shim pins &p[0] */
    memcpy(c, tok, strlen(tok) + 1);
    ptrs[n] = c; lens[n] = (uint32_t)strlen(tok); n++;
  }
  cxg_program* prog = NULL;
  int rc = cxg_program_from_literals(ptrs, lens, n, &prog);
  for (uint32_t i = 0; i < n; i++) free((void*)ptrs[i]);
  if (rc != CXG_OK) { fprintf(stderr, "from_literals: %d %s\n", rc, cxg_last_error()); exit(1); }
  printf("# strategy %s, %u literals\n", cxg_strategy_name(cxg_program_strategy(prog)), n);
  return prog;
}

static cxg_program* build_from_charclass(char* ranges) {
  uint8_t m[256];
  memset(m, 0, sizeof m);
  for (char* tok = strtok(ranges, ","); tok; tok = strtok(NULL, ",")) {
    if (strlen(tok) != 3 || tok[1] != '-') { fprintf(stderr, "range must be X-Y: %s\n", tok); exit(2); }
    for (int b = (uint8_t)tok[0]; b <= (uint8_t)tok[2]; b++) m[b] = 1;
  }
  cxg_program* prog = NULL;
  int rc = cxg_program_from_charclass(m, 1, &prog);
  if (rc != CXG_OK) { fprintf(stderr, "from_charclass: %d %s\n", rc, cxg_last_error()); exit(1); }
  printf("# strategy %s\n", cxg_strategy_name(cxg_program_strategy(prog)));
  return prog;
}

int main(int argc, char** argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s nfa|submatch|literals|charclass SPEC FILE\n", argv[0]); return 2; }
  const char* mode = argv[1];
  uint64_t len = 0;
  uint8_t* hay = read_file(argv[3], &len);
  cxg_program* prog;
  if (!strcmp(mode, "nfa") || !strcmp(mode, "submatch")) prog = build_from_engine(argv[2]);
  else if (!strcmp(mode, "literals")) prog = build_from_literals(argv[2]);
  else if (!strcmp(mode, "charclass")) prog = build_from_charclass(argv[2]);
  else { fprintf(stderr, "unknown mode %s\n", mode); return 2; }
  const int sub = !strcmp(mode, "submatch");
  if (sub ? !cxg_program_submatch_supported(prog) : !cxg_program_supported(prog)) {
    fprintf(stderr, "device path refuses the program: %s\n", cxg_last_error());   /* e.hip stays nil */
    cxg_program_destroy(prog);
    return 4;
  }
  const uint64_t width = sub ? 2u * (uint64_t)cxg_program_num_groups(prog) : 2u;

  /* findAllHip: results := make([][2]int, 0, len(haystack)/100+1); retry on CXG_E_CAPACITY */
  uint64_t cap = len / 100 + 1;
  int64_t* rows = (int64_t*)malloc(cap * width * sizeof(int64_t));
  uint64_t got = 0;
  int retries = 0;
  for (;;) {
    int rc = sub ? cxg_find_all_submatch(prog, hay, len, -1, rows, cap, &got) : cxg_find_all(prog, hay, len, -1, rows, cap, &got);
    if (rc == CXG_OK) break;
    if (rc == CXG_E_CAPACITY) {                      /* got = rows required: grow and retry */
      cap = got;
      rows = (int64_t*)realloc(rows, cap * width * sizeof(int64_t));
      retries++;
      continue;
    }
    fprintf(stderr, "find_all: %d %s\n", rc, cxg_last_error());   /* degrade: the Go caller runs its CPU loop */
    return 1;
  }
  printf("# %llu rows, %d capacity retries\n", (unsigned long long)got, retries);
  for (uint64_t i = 0; i < got; i++) {
    for (uint64_t k = 0; k < width; k++) printf(k ? " %lld" : "%lld", (long long)rows[i * width + k]);
    printf("\n");
  }
  uint64_t cnt = 0;
  if (!sub && (cxg_count(prog, hay, len, -1, &cnt) != CXG_OK || cnt != got)) { fprintf(stderr, "count mismatch\n"); return 1; }
  free(rows);
  free(hay);
  cxg_program_destroy(prog);
  cxg_thread_release();
  return 0;
}
