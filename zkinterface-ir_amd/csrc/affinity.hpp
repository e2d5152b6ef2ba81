// A helper thread next to the thread it works for.
//
// The decoder thread hands the recorder arrays of gates that the recorder frees and the decoder's next message reuses:
// producer and consumer of the same cache lines.  On the MI355X boxes (2 x EPYC 9575F: 16 CCDs of 8 cores, an L3 each) a
// decoder on ANOTHER CCD than the recorder decodes the 1.3 GB C4 relation in 0.7-1.0 s instead of 0.28 s -- every line it
// writes is held by the other CCD's L3 -- and ingest takes 0.9 s instead of 0.41, slower than without the helper (0.63); on
// the same CCD, or its SMT siblings, 0.41 s (profiles/r04_tuning_sweeps.txt, "ingest": taskset matrix).  The decoder
// therefore FOLLOWS its consumer: the consumer notes the CPU it runs on each time it takes a message, and the decoder
// confines itself to the CPUs that share the last-level cache with that CPU (sysfs cache/index3/shared_cpu_list,
// intersected with the process's own mask) whenever that set changes.  Left alone when the set cannot be read, has fewer
// than two CPUs, or ZKI_THREAD_AFFINITY=0.  The consumer's own affinity is never touched; the scheduling worker and the
// task pool only read what the recorder appends (no line goes back and forth) and are placed by the kernel.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>

namespace zki {

struct NearbyCpus {
  cpu_set_t set;
  bool valid = false;
};

inline NearbyCpus cpus_near(int cpu) {
  NearbyCpus out;
  CPU_ZERO(&out.set);
  const char* env = getenv("ZKI_THREAD_AFFINITY");
  if (env && env[0] == '0') return out;
  if (cpu < 0) return out;
  char path[160], list[4096];
  bool have = false;
  snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
  if (FILE* f = fopen(path, "r")) {
    have = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
  }
  // no L3 entry: the caller's NUMA node (/sys/devices/system/cpu/cpu<N>/node<M> exists for exactly one M)
  for (int m = 0; m < 64 && !have; ++m) {
    snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/node%d/cpulist", cpu, m);
    FILE* f = fopen(path, "r");
    if (!f) continue;
    have = fgets(list, sizeof list, f) != nullptr;
    fclose(f);
  }
  if (!have) return out;
  cpu_set_t allowed;
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return out;
  int count = 0;
  for (char* p = list; *p;) {   // "0-63,128-191"
    char* end;
    const long a = strtol(p, &end, 10);
    if (end == p) break;
    long b = a;
    p = end;
    if (*p == '-') {
      b = strtol(p + 1, &end, 10);
      p = end;
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
      if (c >= 0 && CPU_ISSET(c, &allowed)) {
        CPU_SET(c, &out.set);
        ++count;
      }
    while (*p == ',' || *p == ' ' || *p == '\n') ++p;
  }
  out.valid = count >= 2;
  return out;
}

inline NearbyCpus cpus_near_caller() { return cpus_near(sched_getcpu()); }

inline void keep_near(std::thread& t, const NearbyCpus& near) {
  if (near.valid && t.joinable()) pthread_setaffinity_np(t.native_handle(), sizeof near.set, &near.set);
}

// the calling thread moves to the CPUs around `cpu` unless it is among them already; remembers the last set it chose
class FollowCpu {
 public:
  void follow(int cpu) {
    if (cpu < 0 || (have_ && CPU_ISSET(cpu, &current_.set))) return;
    const NearbyCpus near = cpus_near(cpu);
    if (!near.valid) return;
    if (pthread_setaffinity_np(pthread_self(), sizeof near.set, &near.set) == 0) {
      current_ = near;
      have_ = true;
    }
  }

 private:
  NearbyCpus current_;
  bool have_ = false;
};

}  // namespace zki
