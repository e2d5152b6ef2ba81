// Arbitrary-precision unsigned integers for the host-side consumers that must
// accept any field the IR can name (the Validator's range and primality checks,
// rust/src/consumers/validator.rs:176-184,804-819; rust/src/structs/value.rs:52-55).
// The reference uses num-bigint / num-bigint-dig; only the operations those call
// sites need are provided.  Device arithmetic does not use this type.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "structs.hpp"

namespace zki {

class BigNat {
 public:
  BigNat() {}
  explicit BigNat(uint64_t v);
  static BigNat from_bytes_le(const Value& v);  // BigUint::from_bytes_le

  bool is_zero() const { return w_.empty(); }
  bool is_even() const { return w_.empty() || (w_[0] & 1) == 0; }
  size_t bits() const;
  int cmp(const BigNat& o) const;  // -1, 0, 1
  bool operator==(const BigNat& o) const { return cmp(o) == 0; }
  bool operator!=(const BigNat& o) const { return cmp(o) != 0; }
  bool operator<(const BigNat& o) const { return cmp(o) < 0; }
  bool operator>=(const BigNat& o) const { return cmp(o) >= 0; }

  BigNat add(const BigNat& o) const;
  BigNat sub(const BigNat& o) const;  // requires *this >= o
  BigNat mul(const BigNat& o) const;
  BigNat mod(const BigNat& m) const;  // m != 0
  BigNat shr(size_t n) const;
  uint32_t mod_small(uint32_t d) const;
  BigNat powmod(const BigNat& e, const BigNat& m) const;

  std::string to_decimal() const;  // Display of BigUint

 private:
  void trim();
  std::vector<uint32_t> w_;  // little-endian 32-bit limbs, no leading zeros
};

// is_probably_prime (structs/value.rs:52-55 -> num_bigint_dig::prime::probably_prime(n, 10)).
// Both are probabilistic tests; this one is trial division by the primes below 1000, Miller-Rabin to 24 fixed
// prime bases (exact below 3.3e24) and a strong Lucas test with Selfridge's parameters (so it contains
// Baillie-PSW, for which no pseudoprime is known); it agrees with the reference on every modulus that was not
// constructed to defeat one of the two tests.
bool is_probably_prime(const Value& v);
// the Lucas half on its own (n odd and > 2), exposed for its unit test
bool strong_lucas_selfridge(const BigNat& n);

}  // namespace zki
