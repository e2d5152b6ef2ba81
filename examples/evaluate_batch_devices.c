/* One process, several GPUs: a batch of statements over the same relation, split over the devices of the node.
 *
 * The statement of the workspace (its Instance / Witness messages) is replicated into a batch of B lanes -- lane 1 gets
 * its first instance value damaged, so one statement of the batch is false unless that value is unconstrained -- and the
 * batch is evaluated across every visible GPU (or the devices given with --devices): the lanes are split into
 * contiguous shares, one engine and one host thread per device, and zkgpu_counts combines the per-device
 * {satisfied, failed} counters with one RCCL all-reduce (by a host sum when a device is listed twice: "--devices 0,0"
 * rehearses the lane split on a one-GPU box).  What rust/src/cli.rs:315-320 does for one statement, for B of them.
 *
 *   gcc -std=c99 -Iinclude examples/evaluate_batch_devices.c -Lzkinterface-ir_amd/lib -lzkgpu \
 *       -Wl,-rpath,$PWD/zkinterface-ir_amd/lib -o evaluate_batch_devices
 *   ./evaluate_batch_devices [--devices 0,1,...] [--batch B] <workspace dir | file.sieve ...>
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zkgpu.h"

int main(int argc, char** argv) {
  const char* devices = NULL;
  unsigned batch = 1024;
  int first = 1;
  while (first + 1 < argc && argv[first][0] == '-') {
    if (strcmp(argv[first], "--devices") == 0) devices = argv[first + 1];
    else if (strcmp(argv[first], "--batch") == 0) batch = (unsigned)atoi(argv[first + 1]);
    else break;
    first += 2;
  }
  if (argc <= first || batch == 0) {
    fprintf(stderr, "usage: %s [--devices 0,1,...] [--batch B] <paths...>\n", argv[0]);
    return 2;
  }
  char list[512] = "";
  if (!devices) { /* every GPU the runtime sees */
    int n = zkgpu_device_count();
    if (n <= 0) {
      fprintf(stderr, "no GPU visible\n");
      return 2;
    }
    for (int k = 0; k < n && strlen(list) + 8 < sizeof list; ++k) sprintf(list + strlen(list), k ? ",%d" : "%d", k);
    devices = list;
  }
  zkgpu_session* s = zkgpu_session_new();
  if (!s) return 2;
  if (zkgpu_set_option(s, "devices", devices) != 0 ||
      zkgpu_ingest_paths(s, (const char* const*)(argv + first), (size_t)(argc - first)) != 0 || zkgpu_finalize(s, 0) != 0) {
    fprintf(stderr, "setup failed: %s\n", zkgpu_last_error(s));
    return 2;
  }
  /* the workspace's own values, one row per lane */
  const unsigned w = zkgpu_elem_bytes(s), ni = zkgpu_n_instance(s), nw = zkgpu_n_witness(s);
  unsigned char* inst = (unsigned char*)calloc((size_t)batch * ni * w + 1, 1);
  unsigned char* wit = (unsigned char*)calloc((size_t)batch * nw * w + 1, 1);
  if (!inst || !wit) return 2;
  for (unsigned k = 0; k < ni; ++k) zkgpu_message_value(s, 0, k, inst + (size_t)k * w, w);
  for (unsigned k = 0; k < nw; ++k) zkgpu_message_value(s, 1, k, wit + (size_t)k * w, w);
  for (unsigned lane = 1; lane < batch; ++lane) {
    memcpy(inst + (size_t)lane * ni * w, inst, (size_t)ni * w);
    memcpy(wit + (size_t)lane * nw * w, wit, (size_t)nw * w);
  }
  if (batch > 1 && ni) inst[(size_t)1 * ni * w] ^= 1; /* lane 1: first instance value off by one bit */
  uint64_t counts[2] = {0, 0};
  if (zkgpu_set_inputs(s, inst, wit, batch) != 0 || zkgpu_replay(s) != 0 || zkgpu_synchronize(s) != 0 ||
      zkgpu_counts(s, counts) != 0) {
    fprintf(stderr, "replay failed: %s\n", zkgpu_last_error(s));
    return 2;
  }
  char text[1024];
  printf("%u statements over %d engine(s) [devices %s]: %llu TRUE, %llu FALSE\n", batch, zkgpu_n_engines(s), devices,
         (unsigned long long)counts[0], (unsigned long long)counts[1]);
  for (unsigned lane = 0; lane < batch && lane < 3; ++lane) {
    size_t n = zkgpu_lane_violations(s, lane, text, sizeof text);
    printf("lane %u: %s%s\n", lane, n ? "FALSE: " : "TRUE", n ? text : "");
  }
  free(inst);
  free(wit);
  zkgpu_session_free(s);
  return counts[0] + counts[1] == batch ? 0 : 1;
}
