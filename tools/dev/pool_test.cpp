// How long a parallel section of the scheduler's TaskPool takes on this host: three tasks of 500 us each, 40 sections
// 200 us apart (what ordering the runs of a GF(2) level looks like).  1500 = serial, 500 = all three at once.
//   sed -n '/^class TaskPool/,/^};/p' zkinterface-ir_amd/csrc/schedule.cpp > /tmp/pool.inc && g++ -O2 -pthread -o /tmp/pool_test tools/dev/pool_test.cpp
#include <stdint.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "/tmp/pool.inc"
int main() {
  TaskPool p(7);
  for (int rep = 0; rep < 40; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    const std::function<void(uint32_t)> f = [&](uint32_t) {
      auto a = std::chrono::steady_clock::now();
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() < 500e-6) {}
    };
    p.run(3, f);
    printf("%.0f ", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6);
    auto a = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count() < 200e-6) {}
  }
  printf("\n");
}
