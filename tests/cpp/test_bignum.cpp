// Unit test of csrc/sieve/bignum.cpp: the strong Lucas-Selfridge test against its published behaviour
// (every odd prime passes; the composites that pass below 60000 are exactly the known strong Lucas
// pseudoprimes, OEIS A217255), and is_probably_prime() on the same range against a sieve.
#include <stdio.h>

#include <set>
#include <vector>

#include "sieve/bignum.hpp"

using namespace zki;

int main() {
  const int N = 60000;
  std::vector<bool> composite(N + 1, false);
  for (int i = 2; i * i <= N; ++i)
    if (!composite[i])
      for (int j = i * i; j <= N; j += i) composite[j] = true;
  const std::set<int> slpsp = {5459, 5777, 10877, 16109, 18971, 22499, 24569, 25199, 40309, 58519};
  int bad = 0;
  for (int n = 3; n <= N; n += 2) {
    const bool lucas = strong_lucas_selfridge(BigNat((uint64_t)n));
    const bool want = !composite[n] || slpsp.count(n);
    if (lucas != want) {
      printf("strong Lucas(%d) = %d, expected %d\n", n, lucas, want);
      ++bad;
    }
  }
  for (int n = 0; n <= N; ++n) {
    Value v;
    for (int k = n; k; k >>= 8) v.push_back((uint8_t)k);
    const bool p = is_probably_prime(v);
    if (p != (n >= 2 && !composite[n])) {
      printf("is_probably_prime(%d) = %d\n", n, p);
      ++bad;
    }
  }
  printf("bad=%d\n", bad);
  return bad != 0;
}
