/* Minimal C caller of libzkgpu.so: `zki_sieve evaluate <workspace>` (or, with --valid-eval-metrics, the three-part
 * report of `zki_sieve valid-eval-metrics`, rust/src/cli.rs:333-363) for one statement.
 *
 *   gcc -std=c99 -Iinclude examples/evaluate_workspace.c -Lzkinterface-ir_amd/lib -lzkgpu \
 *       -Wl,-rpath,$PWD/zkinterface-ir_amd/lib -o evaluate_workspace
 *   ./evaluate_workspace [--record-only | --valid-eval-metrics] <workspace dir | file.sieve ...>
 *
 * --record-only stops after recording + scheduling (no GPU needed) and prints the tape facts. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkgpu.h"

int main(int argc, char** argv) {
  int record_only = argc > 1 && strcmp(argv[1], "--record-only") == 0;
  int vem = argc > 1 && strcmp(argv[1], "--valid-eval-metrics") == 0;
  int first = (record_only || vem) ? 2 : 1;
  if (argc <= first) {
    fprintf(stderr, "usage: %s [--record-only | --valid-eval-metrics] <paths...>\n", argv[0]);
    return 2;
  }
  zkgpu_session* s = zkgpu_session_new();
  if (!s) return 2;
  /* one statement, one witness: schedule the tape window by window while the relation is still being read */
  if (!record_only) zkgpu_set_option(s, "stream", "1");
  if (vem) { /* the Validator (as prover) and the Stats see every message next to the Evaluator */
    zkgpu_set_option(s, "validate", "prover");
    zkgpu_set_option(s, "metrics", "1");
  }
  if (zkgpu_ingest_paths(s, (const char* const*)(argv + first), (size_t)(argc - first)) != 0) {
    fprintf(stderr, "ingest failed: %s\n", zkgpu_last_error(s));
    return 2;
  }
  char text[4096];
  int invalid = 0;
  if (vem) {
    invalid = zkgpu_validator_count(s) > 0;
    if (invalid) {
      zkgpu_validator_violations(s, text, sizeof text);
      fprintf(stderr, "\nThe statement is NOT COMPLIANT with the specification!\nViolations:\n- %s\n\n", text);
    } else {
      fprintf(stderr, "\nThe statement is COMPLIANT with the specification!\n");
    }
  }
  if (zkgpu_finalize(s, 0) != 0) { /* no relation reached the backend */
    zkgpu_host_violations(s, text, sizeof text);
    fprintf(stderr, "\nThe statement is NOT TRUE!\nViolations:\n- %s\n\n", text);
    zkgpu_session_free(s);
    return 1;
  }
  uint64_t info[8];
  zkgpu_schedule_info(s, info);
  printf("backend calls %llu (asserts %llu), levels %llu, launches %llu, wire-table slots %llu\n",
         (unsigned long long)zkgpu_tape_len(s), (unsigned long long)zkgpu_tape_asserts(s),
         (unsigned long long)info[0], (unsigned long long)info[1], (unsigned long long)info[2]);
  if (record_only) {
    zkgpu_session_free(s);
    return 0;
  }
  if (zkgpu_set_inputs_from_messages(s) != 0 || zkgpu_replay(s) != 0 || zkgpu_synchronize(s) != 0) {
    fprintf(stderr, "replay failed: %s\n", zkgpu_last_error(s));
    zkgpu_session_free(s);
    return 2;
  }
  size_t n = zkgpu_lane_violations(s, 0, text, sizeof text);
  if (n) fprintf(stderr, "\nThe statement is NOT TRUE!\nViolations:\n- %s\n\n", text);
  else fprintf(stderr, "\nThe statement is TRUE!\n");
  if (vem) { /* Stats as serde_json's pretty text, on stdout like the reference */
    size_t len = zkgpu_stats_json(s, NULL, 0);
    char* json = (char*)malloc(len + 1);
    if (json) {
      zkgpu_stats_json(s, json, len + 1);
      printf("%s\n", json);
      free(json);
    }
  }
  zkgpu_session_free(s);
  return (n || invalid) ? 1 : 0;
}
