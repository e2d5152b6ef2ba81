// TEST INFRASTRUCTURE ONLY (oracle/): never linked into, imported by or called
// from the product path.  See oracle/README.md.
//
// Arbitrary-precision unsigned integer standing in for num-bigint 0.3.0's
// `BigUint` (rust/Cargo.lock:202-203), which the reference's PlaintextBackend
// uses for every wire value (rust/src/consumers/evaluator.rs:858-947).
// Heap-allocated little-endian 64-bit digits, schoolbook multiply, Knuth
// algorithm D division -- deliberately NOT Montgomery/Barrett, so that it is
// independent of the arithmetic used by the HIP kernels.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include <algorithm>
#include <stdexcept>

namespace zko {

typedef unsigned __int128 u128;

struct BigUint {
  std::vector<uint64_t> d;  // little-endian, no trailing zero digits

  BigUint() {}
  explicit BigUint(uint64_t v) { if (v) d.push_back(v); }

  void trim() { while (!d.empty() && d.back() == 0) d.pop_back(); }
  bool is_zero() const { return d.empty(); }
  bool is_one() const { return d.size() == 1 && d[0] == 1; }
  bool is_odd() const { return !d.empty() && (d[0] & 1); }

  // BigUint::from_bytes_le: any length, no reduction (evaluator.rs:862-864).
  static BigUint from_bytes_le(const uint8_t* p, size_t n) {
    BigUint r;
    r.d.assign((n + 7) / 8, 0);
    for (size_t i = 0; i < n; ++i) r.d[i / 8] |= (uint64_t)p[i] << (8 * (i % 8));
    r.trim();
    return r;
  }
  static BigUint from_bytes_le(const std::vector<uint8_t>& v) { return from_bytes_le(v.data(), v.size()); }

  // Fixed-width little-endian export; returns false if the value does not fit.
  bool to_bytes_le(uint8_t* out, size_t n) const {
    std::fill(out, out + n, 0);
    for (size_t i = 0; i < d.size() * 8; ++i) {
      uint8_t b = (uint8_t)(d[i / 8] >> (8 * (i % 8)));
      if (i < n) out[i] = b;
      else if (b) return false;
    }
    return true;
  }

  static int cmp(const BigUint& a, const BigUint& b) {
    if (a.d.size() != b.d.size()) return a.d.size() < b.d.size() ? -1 : 1;
    for (size_t i = a.d.size(); i-- > 0;)
      if (a.d[i] != b.d[i]) return a.d[i] < b.d[i] ? -1 : 1;
    return 0;
  }
  bool operator==(const BigUint& o) const { return d == o.d; }
  bool operator<(const BigUint& o) const { return cmp(*this, o) < 0; }

  static BigUint add(const BigUint& a, const BigUint& b) {
    const BigUint& x = a.d.size() >= b.d.size() ? a : b;
    const BigUint& y = a.d.size() >= b.d.size() ? b : a;
    BigUint r;
    r.d.resize(x.d.size() + 1);
    u128 c = 0;
    for (size_t i = 0; i < x.d.size(); ++i) {
      c += (u128)x.d[i] + (i < y.d.size() ? y.d[i] : 0);
      r.d[i] = (uint64_t)c;
      c >>= 64;
    }
    r.d[x.d.size()] = (uint64_t)c;
    r.trim();
    return r;
  }

  // a - b, requires a >= b
  static BigUint sub(const BigUint& a, const BigUint& b) {
    if (cmp(a, b) < 0) throw std::runtime_error("panic: BigUint subtraction underflow");
    BigUint r;
    r.d.resize(a.d.size());
    uint64_t borrow = 0;
    for (size_t i = 0; i < a.d.size(); ++i) {
      u128 t = (u128)a.d[i] - (i < b.d.size() ? b.d[i] : 0) - borrow;
      r.d[i] = (uint64_t)t;
      borrow = (uint64_t)(t >> 64) & 1;
    }
    r.trim();
    return r;
  }

  static BigUint mul(const BigUint& a, const BigUint& b) {
    BigUint r;
    if (a.is_zero() || b.is_zero()) return r;
    r.d.assign(a.d.size() + b.d.size(), 0);
    for (size_t i = 0; i < a.d.size(); ++i) {
      u128 c = 0;
      for (size_t j = 0; j < b.d.size(); ++j) {
        c += (u128)a.d[i] * b.d[j] + r.d[i + j];
        r.d[i + j] = (uint64_t)c;
        c >>= 64;
      }
      r.d[i + b.d.size()] = (uint64_t)c;
    }
    r.trim();
    return r;
  }

  // Knuth TAOCP vol.2 4.3.1 algorithm D; returns remainder, optionally quotient.
  static BigUint rem(const BigUint& u, const BigUint& v, BigUint* quot = nullptr) {
    if (v.is_zero()) throw std::runtime_error("panic: attempt to divide by zero");
    if (cmp(u, v) < 0) {
      if (quot) quot->d.clear();
      return u;
    }
    const size_t n = v.d.size(), m = u.d.size() - n;
    if (n == 1) {
      u128 r = 0;
      BigUint q;
      q.d.resize(u.d.size());
      for (size_t i = u.d.size(); i-- > 0;) {
        u128 cur = (r << 64) | u.d[i];
        q.d[i] = (uint64_t)(cur / v.d[0]);
        r = cur % v.d[0];
      }
      q.trim();
      if (quot) *quot = q;
      return BigUint((uint64_t)r);
    }
    const int s = __builtin_clzll(v.d[n - 1]);
    std::vector<uint64_t> vn(n), un(u.d.size() + 1);
    for (size_t i = n - 1; i > 0; --i) vn[i] = (v.d[i] << s) | (s ? v.d[i - 1] >> (64 - s) : 0);
    vn[0] = v.d[0] << s;
    un[u.d.size()] = s ? u.d[u.d.size() - 1] >> (64 - s) : 0;
    for (size_t i = u.d.size() - 1; i > 0; --i) un[i] = (u.d[i] << s) | (s ? u.d[i - 1] >> (64 - s) : 0);
    un[0] = u.d[0] << s;
    BigUint q;
    q.d.assign(m + 1, 0);
    for (size_t j = m + 1; j-- > 0;) {
      u128 num = ((u128)un[j + n] << 64) | un[j + n - 1];
      u128 qhat = num / vn[n - 1];
      u128 rhat = num % vn[n - 1];
      while ((qhat >> 64) != 0 || (uint64_t)qhat * (u128)vn[n - 2] > ((rhat << 64) | un[j + n - 2])) {
        qhat -= 1;
        rhat += vn[n - 1];
        if ((rhat >> 64) != 0) break;
      }
      // multiply and subtract
      u128 borrow = 0, carry = 0;
      for (size_t i = 0; i < n; ++i) {
        u128 p = (uint64_t)qhat * (u128)vn[i] + carry;
        carry = p >> 64;
        u128 t = (u128)un[i + j] - (uint64_t)p - borrow;
        un[i + j] = (uint64_t)t;
        borrow = (t >> 64) & 1;
      }
      u128 t = (u128)un[j + n] - carry - borrow;
      un[j + n] = (uint64_t)t;
      q.d[j] = (uint64_t)qhat;
      if ((t >> 64) & 1) {  // add back
        q.d[j] -= 1;
        u128 c = 0;
        for (size_t i = 0; i < n; ++i) {
          c += (u128)un[i + j] + vn[i];
          un[i + j] = (uint64_t)c;
          c >>= 64;
        }
        un[j + n] += (uint64_t)c;
      }
    }
    BigUint r;
    r.d.resize(n);
    for (size_t i = 0; i < n; ++i) r.d[i] = (un[i] >> s) | (s && i + 1 < un.size() ? un[i + 1] << (64 - s) : 0);
    r.trim();
    q.trim();
    if (quot) *quot = q;
    return r;
  }

  static BigUint bitand_(const BigUint& a, const BigUint& b) {
    BigUint r;
    r.d.resize(std::min(a.d.size(), b.d.size()));
    for (size_t i = 0; i < r.d.size(); ++i) r.d[i] = a.d[i] & b.d[i];
    r.trim();
    return r;
  }
  static BigUint bitxor_(const BigUint& a, const BigUint& b) {
    BigUint r;
    r.d.resize(std::max(a.d.size(), b.d.size()));
    for (size_t i = 0; i < r.d.size(); ++i)
      r.d[i] = (i < a.d.size() ? a.d[i] : 0) ^ (i < b.d.size() ? b.d[i] : 0);
    r.trim();
    return r;
  }
  BigUint shr1() const {
    BigUint r;
    r.d.resize(d.size());
    for (size_t i = 0; i < d.size(); ++i) r.d[i] = (d[i] >> 1) | (i + 1 < d.size() ? d[i + 1] << 63 : 0);
    r.trim();
    return r;
  }

  std::string to_dec() const {
    if (is_zero()) return "0";
    std::vector<uint64_t> t(d);
    std::string s;
    while (!t.empty()) {
      u128 r = 0;
      for (size_t i = t.size(); i-- > 0;) {
        u128 cur = (r << 64) | t[i];
        t[i] = (uint64_t)(cur / 10000000000000000000ull);
        r = cur % 10000000000000000000ull;
      }
      while (!t.empty() && t.back() == 0) t.pop_back();
      uint64_t chunk = (uint64_t)r;
      for (int k = 0; k < 19; ++k) {
        s.push_back((char)('0' + chunk % 10));
        chunk /= 10;
        if (t.empty() && chunk == 0) break;
      }
    }
    while (s.size() > 1 && s.back() == '0') s.pop_back();
    std::reverse(s.begin(), s.end());
    return s;
  }
};

}  // namespace zko
