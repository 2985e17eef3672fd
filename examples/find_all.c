/* find_all.c — the C ABI from plain C (what a cgo / JNI / ctypes shim binds).
 *
 *   gcc -std=c99 -Iinclude examples/find_all.c -Lcoregex_amd -lcoregex_hip_rocm -o find_all
 *   ./find_all '\d+\.\d+\.\d+\.\d+' access.log
 *
 * Prints every match span as "start end" (absolute byte offsets, Go FindAllIndex semantics).  Needs an MI355X:
 * the library has no CPU search path and reports CXG_E_NO_GPU otherwise.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "coregex_hip.h"

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s PATTERN FILE\n", argv[0]); return 2; }
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 2; }
  fseek(f, 0, SEEK_END);
  long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* hay = (uint8_t*)malloc(n > 0 ? (size_t)n : 1);
  if (fread(hay, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read\n"); return 2; }
  fclose(f);

  cxg_program* prog = NULL;
  int rc = cxg_compile(argv[1], strlen(argv[1]), &prog);
  if (rc != CXG_OK) { fprintf(stderr, "compile: %d %s\n", rc, cxg_last_error()); return 1; }
  printf("# strategy %s, %d group(s), device path %s\n", cxg_strategy_name(cxg_program_strategy(prog)),
         cxg_program_num_groups(prog), cxg_program_supported(prog) ? "yes" : "no (caller keeps its CPU loop)");

  uint64_t got = 0;
  rc = cxg_count(prog, hay, (uint64_t)n, -1, &got);              /* Engine.Count: size the result first */
  if (rc != CXG_OK) { fprintf(stderr, "count: %d %s\n", rc, cxg_last_error()); cxg_program_destroy(prog); return 1; }
  int64_t* spans = (int64_t*)malloc((got ? got : 1) * 2 * sizeof(int64_t));
  rc = cxg_find_all(prog, hay, (uint64_t)n, -1, spans, got, &got);
  if (rc != CXG_OK) { fprintf(stderr, "find_all: %d %s\n", rc, cxg_last_error()); cxg_program_destroy(prog); return 1; }
  for (uint64_t i = 0; i < got; i++) printf("%lld %lld\n", (long long)spans[2 * i], (long long)spans[2 * i + 1]);
  free(spans);
  free(hay);
  cxg_program_destroy(prog);
  return 0;
}
