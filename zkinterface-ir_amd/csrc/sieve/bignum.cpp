#include "bignum.hpp"

#include <algorithm>

namespace zki {

BigNat::BigNat(uint64_t v) {
  while (v) {
    w_.push_back((uint32_t)v);
    v >>= 32;
  }
}

void BigNat::trim() {
  while (!w_.empty() && w_.back() == 0) w_.pop_back();
}

BigNat BigNat::from_bytes_le(const Value& v) {
  BigNat r;
  r.w_.assign((v.size() + 3) / 4, 0);
  for (size_t i = 0; i < v.size(); ++i) r.w_[i / 4] |= (uint32_t)v[i] << (8 * (i % 4));
  r.trim();
  return r;
}

size_t BigNat::bits() const {
  if (w_.empty()) return 0;
  return 32 * (w_.size() - 1) + (32 - __builtin_clz(w_.back()));
}

int BigNat::cmp(const BigNat& o) const {
  if (w_.size() != o.w_.size()) return w_.size() < o.w_.size() ? -1 : 1;
  for (size_t i = w_.size(); i-- > 0;)
    if (w_[i] != o.w_[i]) return w_[i] < o.w_[i] ? -1 : 1;
  return 0;
}

BigNat BigNat::add(const BigNat& o) const {
  BigNat r;
  const size_t n = std::max(w_.size(), o.w_.size());
  r.w_.resize(n + 1);
  uint64_t c = 0;
  for (size_t i = 0; i < n; ++i) {
    c += (uint64_t)(i < w_.size() ? w_[i] : 0) + (i < o.w_.size() ? o.w_[i] : 0);
    r.w_[i] = (uint32_t)c;
    c >>= 32;
  }
  r.w_[n] = (uint32_t)c;
  r.trim();
  return r;
}

BigNat BigNat::sub(const BigNat& o) const {
  BigNat r;
  r.w_.resize(w_.size());
  int64_t borrow = 0;
  for (size_t i = 0; i < w_.size(); ++i) {
    int64_t d = (int64_t)w_[i] - (i < o.w_.size() ? o.w_[i] : 0) - borrow;
    borrow = d < 0;
    r.w_[i] = (uint32_t)(d + (borrow ? (1ll << 32) : 0));
  }
  r.trim();
  return r;
}

BigNat BigNat::mul(const BigNat& o) const {
  BigNat r;
  if (is_zero() || o.is_zero()) return r;
  r.w_.assign(w_.size() + o.w_.size(), 0);
  for (size_t i = 0; i < w_.size(); ++i) {
    uint64_t c = 0;
    for (size_t j = 0; j < o.w_.size(); ++j) {
      c += (uint64_t)w_[i] * o.w_[j] + r.w_[i + j];
      r.w_[i + j] = (uint32_t)c;
      c >>= 32;
    }
    r.w_[i + o.w_.size()] = (uint32_t)c;
  }
  r.trim();
  return r;
}

BigNat BigNat::shr(size_t n) const {
  BigNat r;
  const size_t ws = n / 32, bs = n % 32;
  if (ws >= w_.size()) return r;
  r.w_.resize(w_.size() - ws);
  for (size_t i = 0; i < r.w_.size(); ++i) {
    uint64_t v = w_[i + ws];
    if (i + ws + 1 < w_.size()) v |= (uint64_t)w_[i + ws + 1] << 32;
    r.w_[i] = (uint32_t)(v >> bs);
  }
  r.trim();
  return r;
}

uint32_t BigNat::mod_small(uint32_t d) const {
  uint64_t r = 0;
  for (size_t i = w_.size(); i-- > 0;) r = ((r << 32) | w_[i]) % d;
  return (uint32_t)r;
}

// Remainder by schoolbook long division on normalised limbs (TAOCP 4.3.1 algorithm D, remainder only).
BigNat BigNat::mod(const BigNat& m) const {
  if (m.is_zero()) throw Error("division by zero");
  if (cmp(m) < 0) return *this;
  if (m.w_.size() == 1) return BigNat((uint64_t)mod_small(m.w_[0]));
  const int s = __builtin_clz(m.w_.back());
  const size_t n = m.w_.size(), len = w_.size();
  std::vector<uint32_t> v(n), u(len + 1);
  for (size_t i = n; i-- > 0;) v[i] = (m.w_[i] << s) | (s && i ? m.w_[i - 1] >> (32 - s) : 0);
  u[len] = s ? w_[len - 1] >> (32 - s) : 0;
  for (size_t i = len; i-- > 0;) u[i] = (w_[i] << s) | (s && i ? w_[i - 1] >> (32 - s) : 0);
  for (size_t j = len - n + 1; j-- > 0;) {
    const uint64_t num = ((uint64_t)u[j + n] << 32) | u[j + n - 1];
    uint64_t q = num / v[n - 1], r = num % v[n - 1];
    while (q >> 32 || q * v[n - 2] > ((r << 32) | u[j + n - 2])) {
      --q;
      r += v[n - 1];
      if (r >> 32) break;
    }
    int64_t borrow = 0;
    uint64_t carry = 0;
    for (size_t i = 0; i < n; ++i) {
      carry += q * v[i];
      const int64_t t = (int64_t)u[i + j] - borrow - (int64_t)(carry & 0xFFFFFFFFu);
      u[i + j] = (uint32_t)t;
      borrow = t < 0;
      carry >>= 32;
    }
    const int64_t t = (int64_t)u[j + n] - borrow - (int64_t)carry;
    u[j + n] = (uint32_t)t;
    if (t < 0) {  // q was one too large: add the divisor back
      uint64_t c = 0;
      for (size_t i = 0; i < n; ++i) {
        c += (uint64_t)u[i + j] + v[i];
        u[i + j] = (uint32_t)c;
        c >>= 32;
      }
      u[j + n] += (uint32_t)c;
    }
  }
  BigNat r;
  r.w_.resize(n);
  for (size_t i = 0; i < n; ++i) r.w_[i] = (u[i] >> s) | (s && i + 1 <= n ? (uint32_t)((uint64_t)u[i + 1] << (32 - s)) : 0);
  r.trim();
  return r;
}

BigNat BigNat::powmod(const BigNat& e, const BigNat& m) const {
  BigNat result(1), base = mod(m);
  result = result.mod(m);
  const size_t nb = e.bits();
  for (size_t i = 0; i < nb; ++i) {
    if ((e.w_[i / 32] >> (i % 32)) & 1) result = result.mul(base).mod(m);
    if (i + 1 < nb) base = base.mul(base).mod(m);
  }
  return result;
}

std::string BigNat::to_decimal() const {
  if (is_zero()) return "0";
  std::vector<uint32_t> t = w_;
  std::string out;
  while (!t.empty()) {
    uint64_t r = 0;
    for (size_t i = t.size(); i-- > 0;) {
      const uint64_t cur = (r << 32) | t[i];
      t[i] = (uint32_t)(cur / 1000000000u);
      r = cur % 1000000000u;
    }
    while (!t.empty() && t.back() == 0) t.pop_back();
    for (int k = 0; k < 9; ++k) {
      out.push_back((char)('0' + r % 10));
      r /= 10;
      if (t.empty() && r == 0) break;
    }
  }
  std::reverse(out.begin(), out.end());
  return out;
}

namespace {

uint64_t gcd_free_jacobi_u64(uint64_t a, uint64_t n, int* t) {  // Jacobi (a/n), n odd; returns 0 when gcd != 1
  a %= n;
  while (a != 0) {
    while ((a & 1) == 0) {
      a >>= 1;
      const uint64_t r = n & 7;
      if (r == 3 || r == 5) *t = -*t;
    }
    std::swap(a, n);
    if ((a & 3) == 3 && (n & 3) == 3) *t = -*t;
    a %= n;
  }
  return n == 1 ? 1 : 0;
}

// Jacobi symbol (a/n) for a small signed a and an odd n > 1.
int jacobi_small(int64_t a, const BigNat& n) {
  int t = 1;
  const uint32_t n8 = n.mod_small(8);
  uint64_t A = a < 0 ? (uint64_t)(-a) : (uint64_t)a;
  if (a < 0 && (n8 & 3) == 3) t = -t;
  if (A == 0) return 0;
  while ((A & 1) == 0) {
    A >>= 1;
    if (n8 == 3 || n8 == 5) t = -t;
  }
  if (A == 1) return t;
  if ((A & 3) == 3 && (n8 & 3) == 3) t = -t;
  const uint64_t N = n.mod_small((uint32_t)A);  // callers keep |a| below 2^31
  return gcd_free_jacobi_u64(N, A, &t) ? t : 0;
}

bool is_perfect_square(const BigNat& n) {
  // integer square root by bisection: lo^2 <= n < hi^2 throughout
  BigNat lo, hi(1);
  for (size_t i = 0; i < n.bits() / 2 + 1; ++i) hi = hi.add(hi);
  while (BigNat(1) < hi.sub(lo)) {
    const BigNat mid = lo.add(hi).shr(1);
    if (n < mid.mul(mid)) hi = mid;
    else lo = mid;
  }
  return lo.mul(lo) == n;
}

}  // namespace

// Strong Lucas probable-prime test with Selfridge's parameters (the second half of Baillie-PSW).
bool strong_lucas_selfridge(const BigNat& n) {  // n odd, > 2
  if (is_perfect_square(n)) return false;
  int64_t D = 5;
  for (int tries = 0;; ++tries) {
    const int j = jacobi_small(D, n);
    if (j == -1) break;
    if (j == 0 && BigNat((uint64_t)(D < 0 ? -D : D)) < n) return false;  // a small factor
    if (tries > 200) return true;  // no suitable D found: leave the verdict to Miller-Rabin
    D = D > 0 ? -(D + 2) : -(D - 2);
  }
  const BigNat one(1), two(2);
  auto addm = [&](const BigNat& a, const BigNat& b) { BigNat r = a.add(b); return r >= n ? r.sub(n) : r; };
  auto subm = [&](const BigNat& a, const BigNat& b) { return a >= b ? a.sub(b) : a.add(n).sub(b); };
  auto mulm = [&](const BigNat& a, const BigNat& b) { return a.mul(b).mod(n); };
  auto half = [&](const BigNat& a) { return a.is_even() ? a.shr(1) : a.add(n).shr(1); };
  auto from_signed = [&](int64_t v) { return v >= 0 ? BigNat((uint64_t)v).mod(n) : subm(BigNat(), BigNat((uint64_t)(-v)).mod(n)); };
  const BigNat Dm = from_signed(D), Q = from_signed((1 - D) / 4);
  const BigNat n1 = n.add(one);
  size_t s = 0;
  while (n1.shr(s).is_even()) ++s;
  const BigNat d = n1.shr(s);
  // U_1 = 1, V_1 = P = 1, Q^1 = Q; walk the bits of d below the top one
  BigNat U = one, V = one, Qk = Q;
  for (size_t b = d.bits() - 1; b-- > 0;) {
    U = mulm(U, V);
    V = subm(mulm(V, V), addm(Qk, Qk));
    Qk = mulm(Qk, Qk);
    if (!d.shr(b).is_even()) {
      const BigNat U2 = half(addm(U, V));            // (P U + V) / 2, P = 1
      const BigNat V2 = half(addm(mulm(Dm, U), V));  // (D U + P V) / 2
      U = U2;
      V = V2;
      Qk = mulm(Qk, Q);
    }
  }
  if (U.is_zero() || V.is_zero()) return true;
  for (size_t r = 1; r < s; ++r) {
    V = subm(mulm(V, V), addm(Qk, Qk));
    Qk = mulm(Qk, Qk);
    if (V.is_zero()) return true;
  }
  return false;
}

bool is_probably_prime(const Value& v) {
  const BigNat n = BigNat::from_bytes_le(v);
  if (n < BigNat(2)) return false;
  static std::vector<uint32_t> small;
  if (small.empty()) {
    for (uint32_t c = 2; c < 1000; ++c) {
      bool prime = true;
      for (uint32_t d = 2; d * d <= c; ++d)
        if (c % d == 0) prime = false;
      if (prime) small.push_back(c);
    }
  }
  for (uint32_t p : small) {
    if (n == BigNat(p)) return true;
    if (n.mod_small(p) == 0) return false;
  }
  const BigNat one(1), n1 = n.sub(one);
  size_t r = 0;
  while (((n1.shr(r)).is_even())) ++r;
  const BigNat d = n1.shr(r);
  for (size_t k = 0; k < 24; ++k) {
    BigNat x = BigNat(small[k]).powmod(d, n);
    if (x == one || x == n1) continue;
    bool witness = true;
    for (size_t i = 1; i < r && witness; ++i) {
      x = x.mul(x).mod(n);
      if (x == n1) witness = false;
    }
    if (witness) return false;
  }
  return strong_lucas_selfridge(n);
}

}  // namespace zki
