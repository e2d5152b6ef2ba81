// csrc/affinity.hpp: the CPUs around a CPU, and a thread that follows another one.
#include <atomic>
#include <cstdio>
#include <thread>

#include "affinity.hpp"

int main() {
  using namespace zki;
  int bad = 0;
  cpu_set_t allowed;
  if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return 2;
  const int me = sched_getcpu();
  const NearbyCpus near = cpus_near(me);
  int count = 0;
  if (near.valid) {
    // the set holds the CPU it was asked about and nothing the process may not run on
    bad += !CPU_ISSET(me, &near.set);
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &near.set)) {
        ++count;
        bad += !CPU_ISSET(c, &allowed);
      }
    bad += count < 2;
  }
  // an impossible CPU gives no set, and following it does nothing
  bad += cpus_near(-1).valid;
  bad += cpus_near(1 << 20).valid;
  // a thread that follows this one ends up allowed on this thread's CPU (where the sets can be read), and never anywhere
  // the process may not run
  std::atomic<int> where{-2};
  bool inside = true;
  std::thread t([&] {
    FollowCpu place;
    place.follow(-1);
    place.follow(me);
    place.follow(me);   // (in the set already: nothing to do)
    cpu_set_t mine;
    pthread_getaffinity_np(pthread_self(), sizeof mine, &mine);
    for (int c = 0; c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &mine) && !CPU_ISSET(c, &allowed)) inside = false;
    if (near.valid && !CPU_ISSET(me, &mine)) inside = false;
    where.store(sched_getcpu());
  });
  t.join();
  bad += !inside;
  bad += where.load() < 0;
  // the caller's own mask is what it was
  cpu_set_t after;
  sched_getaffinity(0, sizeof after, &after);
  bad += !CPU_EQUAL(&allowed, &after);
  printf("cpu %d, %d cpus around it (%s), bad=%d\n", me, count, near.valid ? "valid" : "not readable here", bad);
  return bad ? 1 : 0;
}
