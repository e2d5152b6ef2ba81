// GF(p) arithmetic for the SIEVE IR tape replay kernels (gfx950).
//
// Replaces, on device, the arithmetic of the reference's PlaintextBackend
// (rust/src/consumers/evaluator.rs:908-922: `(a + b) % m`, `(a * b) % m` on
// num-bigint BigUint).  Wire values live in Montgomery form x*R mod p with
// R = 2^(64*NL); the stored limbs are NL 64-bit little-endian limbs, handled
// here as N = 2*NL 32-bit words because the CDNA4 integer multiplier is
// 32x32 (v_mad_u64_u32).  All results are canonical (< p), which is what makes
// from_mont(x) bit-identical to the reference's reduced BigUint.
//
// Requirements: p odd, p < 2^(32*N), N <= 12 (384 bits).  p = 2 is handled by the Boolean path.
#pragma once
#include "args.hpp"

namespace zkgpu {

template <int N>
struct Fp {
  u32 w[N];
};

template <int N>
__device__ __forceinline__ bool fp_is_zero(const Fp<N>& a) {
  u32 acc = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) acc |= a.w[i];
  return acc == 0;
}

// 1 (Montgomery form) if a != 0, else 0: the value of a^(p-1) over a prime field (Fermat), which is what the
// Switch indicator ladder of the reference computes (evaluator.rs:801-839).
template <int N>
__device__ __forceinline__ Fp<N> fp_nonzero_indicator(const Fp<N>& a, const FieldParams& fp) {
  const bool z = fp_is_zero<N>(a);
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = z ? 0u : fp.one[i];
  return r;
}

// forward declarations for the integer-bitwise gates below
template <int N>
__device__ __forceinline__ Fp<N> fp_from_mont(const Fp<N>& a, const FieldParams& fp);
template <int N>
__device__ __forceinline__ Fp<N> fp_to_mont(const Fp<N>& a, const FieldParams& fp);
template <int N>
__device__ __forceinline__ bool fp_geq_p(const Fp<N>& a, const FieldParams& fp);

// `and` / `xor` of PlaintextBackend over an odd field (evaluator.rs:924-933): the bit operation on the canonical
// integers, then `% p`.  a & b <= min(a, b) < p needs no reduction; a ^ b < 2^bits(p) < 2p needs one subtraction.
template <int N>
__device__ __forceinline__ Fp<N> fp_bit_and(const Fp<N>& a, const Fp<N>& b, const FieldParams& fp) {
  const Fp<N> x = fp_from_mont<N>(a, fp), y = fp_from_mont<N>(b, fp);
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = x.w[i] & y.w[i];
  return fp_to_mont<N>(r, fp);
}
template <int N>
__device__ __forceinline__ Fp<N> fp_bit_xor(const Fp<N>& a, const Fp<N>& b, const FieldParams& fp) {
  const Fp<N> x = fp_from_mont<N>(a, fp), y = fp_from_mont<N>(b, fp);
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = x.w[i] ^ y.w[i];
  if (fp_geq_p<N>(r, fp)) {
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const u64 d = (u64)r.w[i] - fp.p[i] - borrow;
      r.w[i] = (u32)d;
      borrow = (d >> 63) & 1;
    }
  }
  return fp_to_mont<N>(r, fp);
}
// 1 (Montgomery form) or 0: `not` of PlaintextBackend (evaluator.rs:935-938) is fp_indicator(the integer is zero)
template <int N>
__device__ __forceinline__ Fp<N> fp_indicator(bool set, const FieldParams& fp) {
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = set ? fp.one[i] : 0u;
  return r;
}
template <int N>
__device__ __forceinline__ Fp<N> fp_is_zero_indicator(const Fp<N>& a, const FieldParams& fp) {
  return fp_indicator<N>(fp_is_zero<N>(a), fp);
}

// a >= p ?
template <int N>
__device__ __forceinline__ bool fp_geq_p(const Fp<N>& a, const FieldParams& fp) {
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u64 d = (u64)a.w[i] - fp.p[i] - borrow;
    borrow = (d >> 63) & 1;  // high bits set on underflow
  }
  return borrow == 0;
}

// r = (a + b) mod p for canonical a, b.
template <int N>
__device__ __forceinline__ Fp<N> fp_add(const Fp<N>& a, const Fp<N>& b, const FieldParams& fp) {
  Fp<N> s, d;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    c += (u64)a.w[i] + b.w[i];
    s.w[i] = (u32)c;
    c >>= 32;
  }
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u64 t = (u64)s.w[i] - fp.p[i] - borrow;
    d.w[i] = (u32)t;
    borrow = (t >> 63) & 1;
  }
  // s >= p  <=>  carry-out of the add, or no borrow in the subtract
  const bool use_d = (c != 0) | (borrow == 0);
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = use_d ? d.w[i] : s.w[i];
  return r;
}

// (acc_hi : acc_lo) += x * y with a 96-bit accumulator: one v_mad_u64_u32 whose carry-out (VCC) is
// folded into the third word by one v_addc_co_u32.  hipcc's own lowering of the same arithmetic from
// C spends ~5 instructions per word product (zero-extension moves + a 64-bit add).
#define ZKGPU_MADC(lo, hi, x, y)                                                              \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"                \
      : "+v"(lo), "+v"(hi)                                                                    \
      : "v"(x), "v"(y)                                                                        \
      : "vcc")

// The same with the second factor in an SGPR (a wave-uniform coefficient word, a word of p): VOP3 takes one scalar
// source, so no VGPR copy of it is needed.
#define ZKGPU_MADC_S(lo, hi, x, y)                                                            \
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"                \
      : "+v"(lo), "+v"(hi)                                                                    \
      : "v"(x), "s"(y)                                                                        \
      : "vcc")

// A run of K accumulations as ONE asm statement: hipcc's hazard recogniser treats every asm statement as if it could
// be a wide store and puts an `s_nop 0` between two statements that touch the same VGPR -- one per word product when
// each product is a statement of its own, and on gfx940+ one behind EVERY statement that defines a register: up to
// eight products (18 operands) per statement.  madc_run<K, SC>(lo, hi, x, y): (hi : lo) += sum_{j < K} x[j] * y[-j]
// (y walks DOWN, as the operand of a column of a product does); SC: y is wave-uniform and goes in SGPRs.
#define ZKGPU_MT(i, j) "v_mad_u64_u32 %0, vcc, %" #i ", %" #j ", %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
// the first accumulation of a column: the carry word is DEFINED here (0 + 0 + carry), not added to -- see the end of
// mont_reduce_column
#define ZKGPU_MT0(i, j) "v_mad_u64_u32 %0, vcc, %" #i ", %" #j ", %0\n\tv_addc_co_u32 %1, vcc, 0, 0, vcc\n\t"
template <int K, bool SC, bool FIRST = false>
__device__ __forceinline__ void madc_run(u64& lo, u32& hi, const u32* x, const u32* y) {
  static_assert(K >= 1 && K <= 12, "");
#define ZKGPU_IN(j) "v"(x[j]), "v"(y[-(j)])
#define ZKGPU_IS(j) "v"(x[j]), "s"(y[-(j)])
  if constexpr (K == 1) { if constexpr (FIRST) { if constexpr (SC) asm(ZKGPU_MT0(2, 3) : "+v"(lo), "=&v"(hi) : ZKGPU_IS(0) : "vcc"); else asm(ZKGPU_MT0(2, 3) : "+v"(lo), "=&v"(hi) : ZKGPU_IN(0) : "vcc"); } else { if constexpr (SC) asm(ZKGPU_MT(2, 3) : "+v"(lo), "+v"(hi) : ZKGPU_IS(0) : "vcc"); else asm(ZKGPU_MT(2, 3) : "+v"(lo), "+v"(hi) : ZKGPU_IN(0) : "vcc"); } }
  else if constexpr (K == 2) { if constexpr (FIRST) { if constexpr (SC) asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) : "+v"(lo), "=&v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1) : "vcc"); else asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) : "+v"(lo), "=&v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1) : "vcc"); } else { if constexpr (SC) asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) : "+v"(lo), "+v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1) : "vcc"); else asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) : "+v"(lo), "+v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1) : "vcc"); } }
  else if constexpr (K == 3) { if constexpr (FIRST) { if constexpr (SC) asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) : "+v"(lo), "=&v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2) : "vcc"); else asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) : "+v"(lo), "=&v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2) : "vcc"); } else { if constexpr (SC) asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) : "+v"(lo), "+v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2) : "vcc"); else asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) : "+v"(lo), "+v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2) : "vcc"); } }
  else if constexpr (K == 4) { if constexpr (FIRST) { if constexpr (SC) asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) : "+v"(lo), "=&v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2), ZKGPU_IS(3) : "vcc"); else asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) : "+v"(lo), "=&v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2), ZKGPU_IN(3) : "vcc"); } else { if constexpr (SC) asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) : "+v"(lo), "+v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2), ZKGPU_IS(3) : "vcc"); else asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) : "+v"(lo), "+v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2), ZKGPU_IN(3) : "vcc"); } }
  else if constexpr (K == 5) { if constexpr (FIRST) { if constexpr (SC) asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) : "+v"(lo), "=&v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2), ZKGPU_IS(3), ZKGPU_IS(4) : "vcc"); else asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) : "+v"(lo), "=&v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2), ZKGPU_IN(3), ZKGPU_IN(4) : "vcc"); } else { if constexpr (SC) asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) : "+v"(lo), "+v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2), ZKGPU_IS(3), ZKGPU_IS(4) : "vcc"); else asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) : "+v"(lo), "+v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2), ZKGPU_IN(3), ZKGPU_IN(4) : "vcc"); } }
  else if constexpr (K == 6) { if constexpr (FIRST) { if constexpr (SC) asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) : "+v"(lo), "=&v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2), ZKGPU_IS(3), ZKGPU_IS(4), ZKGPU_IS(5) : "vcc"); else asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) : "+v"(lo), "=&v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2), ZKGPU_IN(3), ZKGPU_IN(4), ZKGPU_IN(5) : "vcc"); } else { if constexpr (SC) asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) : "+v"(lo), "+v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2), ZKGPU_IS(3), ZKGPU_IS(4), ZKGPU_IS(5) : "vcc"); else asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) : "+v"(lo), "+v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2), ZKGPU_IN(3), ZKGPU_IN(4), ZKGPU_IN(5) : "vcc"); } }
  else if constexpr (K == 7) { if constexpr (FIRST) { if constexpr (SC) asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) ZKGPU_MT(14, 15) : "+v"(lo), "=&v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2), ZKGPU_IS(3), ZKGPU_IS(4), ZKGPU_IS(5), ZKGPU_IS(6) : "vcc"); else asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) ZKGPU_MT(14, 15) : "+v"(lo), "=&v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2), ZKGPU_IN(3), ZKGPU_IN(4), ZKGPU_IN(5), ZKGPU_IN(6) : "vcc"); } else { if constexpr (SC) asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) ZKGPU_MT(14, 15) : "+v"(lo), "+v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2), ZKGPU_IS(3), ZKGPU_IS(4), ZKGPU_IS(5), ZKGPU_IS(6) : "vcc"); else asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) ZKGPU_MT(14, 15) : "+v"(lo), "+v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2), ZKGPU_IN(3), ZKGPU_IN(4), ZKGPU_IN(5), ZKGPU_IN(6) : "vcc"); } }
  else if constexpr (K == 8) { if constexpr (FIRST) { if constexpr (SC) asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) ZKGPU_MT(14, 15) ZKGPU_MT(16, 17) : "+v"(lo), "=&v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2), ZKGPU_IS(3), ZKGPU_IS(4), ZKGPU_IS(5), ZKGPU_IS(6), ZKGPU_IS(7) : "vcc"); else asm(ZKGPU_MT0(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) ZKGPU_MT(14, 15) ZKGPU_MT(16, 17) : "+v"(lo), "=&v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2), ZKGPU_IN(3), ZKGPU_IN(4), ZKGPU_IN(5), ZKGPU_IN(6), ZKGPU_IN(7) : "vcc"); } else { if constexpr (SC) asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) ZKGPU_MT(14, 15) ZKGPU_MT(16, 17) : "+v"(lo), "+v"(hi) : ZKGPU_IS(0), ZKGPU_IS(1), ZKGPU_IS(2), ZKGPU_IS(3), ZKGPU_IS(4), ZKGPU_IS(5), ZKGPU_IS(6), ZKGPU_IS(7) : "vcc"); else asm(ZKGPU_MT(2, 3) ZKGPU_MT(4, 5) ZKGPU_MT(6, 7) ZKGPU_MT(8, 9) ZKGPU_MT(10, 11) ZKGPU_MT(12, 13) ZKGPU_MT(14, 15) ZKGPU_MT(16, 17) : "+v"(lo), "+v"(hi) : ZKGPU_IN(0), ZKGPU_IN(1), ZKGPU_IN(2), ZKGPU_IN(3), ZKGPU_IN(4), ZKGPU_IN(5), ZKGPU_IN(6), ZKGPU_IN(7) : "vcc"); } }
  else {  // longer runs: eight at a time
    madc_run<8, SC, FIRST>(lo, hi, x, y);
    madc_run<K - 8, SC, false>(lo, hi, x + 8, y - 8);
  }
#undef ZKGPU_IN
#undef ZKGPU_IS
}

// Montgomery product a*b*R^{-1} mod p, product-scanning (column by column) form, canonical out.
// Per column k: acc += sum a[i]*b[k-i] + sum m[i]*p[k-i]; m[k] = acc * n0inv makes the low word 0;
// the accumulator then shifts down one word.  2*N^2 + N word products, 2 instructions each.
// FIRST: these are the first products of column k (every column below 2N - 1 has some): they define the carry word
template <int N, int k, bool SC, bool FIRST>
__device__ __forceinline__ void mont_products(u64& lo, u32& hi, const u32* x, const u32* y) {
  if constexpr (k < N) {
    madc_run<k + 1, SC, FIRST>(lo, hi, x, y + k);                       // x[i] * y[k - i], i = 0..k
  } else if constexpr (k - N + 1 < N) {
    madc_run<2 * N - 1 - k, SC, FIRST>(lo, hi, x + (k - N + 1), y + (N - 1));   // i = k-N+1 .. N-1
  }
}
template <int N, int k>
__device__ __forceinline__ void mont_reduce_column(u64& lo, u32& hi, u32 (&m)[N], u32 (&t)[N + 1], const FieldParams& fp) {
  if constexpr (k < N) {
    if constexpr (k > 0) madc_run<k, true>(lo, hi, &m[0], &fp.p[k]);   // m[i] * p[k - i], i < k: the words of p stay in SGPRs
    m[k] = (u32)lo * fp.n0inv;
    madc_run<1, true>(lo, hi, &m[k], &fp.p[0]);
  } else {
    if constexpr (k - N + 1 < N) madc_run<2 * N - 1 - k, true>(lo, hi, &m[k - N + 1], &fp.p[N - 1]);
    t[k - N] = (u32)lo;
  }
  // The accumulator moves down one word.  Its new top word is not zeroed: the first accumulation of the next column
  // defines it (ZKGPU_MT0), and the carry word of this column becomes the high half of the new pair -- one register
  // move per column (the 64-bit operand of v_mad_u64_u32 has to be an aligned pair) instead of three.
  lo = (lo >> 32) | ((u64)hi << 32);
}

template <int N, int k>
__device__ __forceinline__ void fp_mul_columns(u64& lo, u32& hi, const Fp<N>& a, const Fp<N>& b, u32 (&m)[N], u32 (&t)[N + 1],
                                               const FieldParams& fp) {
  if constexpr (k < 2 * N) {
    mont_products<N, k, false, true>(lo, hi, a.w, b.w);
    mont_reduce_column<N, k>(lo, hi, m, t, fp);
    fp_mul_columns<N, k + 1>(lo, hi, a, b, m, t, fp);
  }
}

template <int N>
__device__ __forceinline__ Fp<N> fp_mul(const Fp<N>& a, const Fp<N>& b, const FieldParams& fp) {
  u64 lo = 0;
  u32 hi = 0;
  u32 m[N], t[N + 1];
  fp_mul_columns<N, 0>(lo, hi, a, b, m, t, fp);
  t[N] = (u32)lo;
  // t < 2p here; one conditional subtraction.
  Fp<N> d;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u64 x = (u64)t[i] - fp.p[i] - borrow;
    d.w[i] = (u32)x;
    borrow = (x >> 63) & 1;
  }
  const bool use_d = (t[N] != 0) | (borrow == 0);
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = use_d ? d.w[i] : t[i];
  return r;
}

// N wave-uniform words (a coefficient in Montgomery form) read on the scalar path into SGPRs
template <int N>
struct FpS {
  u32 w[N];
};
template <int N>
__device__ __forceinline__ FpS<N> fp_load_uniform(const u32* pool, u32 index) {
  typedef const u32 __attribute__((address_space(4))) cu32;
  cu32* q = (cu32*)(unsigned long long)(pool + (size_t)__builtin_amdgcn_readfirstlane(index) * N);
  FpS<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = q[i];
  return r;
}

// sum_k v[k]*c[k]*R^{-1} mod p with ONE Montgomery reduction for the whole sum (lazy reduction): the K
// double-width products are accumulated column by column next to the m*p products of the reduction.
// K*N^2 + N^2 + N word products instead of K*(2*N^2 + N).  The coefficients c[k] are wave-uniform and stay in
// SGPRs, as do the words of p.  The unreduced result is below K*p*p/R + p = (K*p/R + 1)*p: fp.dot_rounds[K-1] =
// ceil(K*p/R) conditional subtractions make it canonical (one for K = 3 over BN254, where p/R = 0.19).
template <int N, int K, int k>
__device__ __forceinline__ void fp_dot_columns(u64& lo, u32& hi, const Fp<N> (&v)[K], const FpS<N> (&c)[K], u32 (&m)[N],
                                               u32 (&t)[N + 1], const FieldParams& fp) {
  if constexpr (k < 2 * N) {
    mont_products<N, k, true, true>(lo, hi, v[0].w, c[0].w);
    if constexpr (K > 1) mont_products<N, k, true, false>(lo, hi, v[1].w, c[1].w);
    if constexpr (K > 2) mont_products<N, k, true, false>(lo, hi, v[2].w, c[2].w);
    if constexpr (K > 3) mont_products<N, k, true, false>(lo, hi, v[3].w, c[3].w);
    mont_reduce_column<N, k>(lo, hi, m, t, fp);
    fp_dot_columns<N, K, k + 1>(lo, hi, v, c, m, t, fp);
  }
}

// `rounds`: conditional subtractions applied to the lazily reduced sum; fp.dot_rounds[K - 1] makes it canonical,
// 0 leaves it below (K * p / R + 1) * p (the caller must know that this fits N words and what may consume it).
template <int N, int K>
__device__ __forceinline__ Fp<N> fp_dot(const Fp<N> (&v)[K], const FpS<N> (&c)[K], const FieldParams& fp, u32 rounds) {
  static_assert(K >= 1 && K <= 4, "the 32-bit carry word of the accumulator holds 2*K*N carries");
  u64 lo = 0;
  u32 hi = 0;
  u32 m[N], t[N + 1];
  fp_dot_columns<N, K, 0>(lo, hi, v, c, m, t, fp);
  t[N] = (u32)lo;
  for (u32 round = 0; round < rounds; ++round) {
    u32 d[N];
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      u64 x = (u64)t[i] - fp.p[i] - borrow;
      d[i] = (u32)x;
      borrow = (x >> 63) & 1;
    }
    const bool ge = (t[N] != 0) | (borrow == 0);  // t >= p
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = ge ? d[i] : t[i];
    t[N] = ge ? t[N] - (u32)borrow : t[N];
  }
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = t[i];
  return r;
}

// Reference form of the same product (CIOS over 32-bit words, plain C): kept for A/B builds
// (-DZKGPU_MUL_PLAIN_C in tools/kbench.hip) and as the readable statement of the arithmetic.
template <int N>
__device__ __forceinline__ Fp<N> fp_mul_plain(const Fp<N>& a, const Fp<N>& b, const FieldParams& fp) {
  u32 t[N + 2];
#pragma unroll
  for (int i = 0; i < N + 2; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u64 c = 0;
    const u32 bi = b.w[i];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      c += (u64)a.w[j] * bi + t[j];
      t[j] = (u32)c;
      c >>= 32;
    }
    c += t[N];
    t[N] = (u32)c;
    t[N + 1] = (u32)(c >> 32);
    const u32 m = t[0] * fp.n0inv;
    c = (u64)m * fp.p[0] + t[0];
    c >>= 32;
#pragma unroll
    for (int j = 1; j < N; ++j) {
      c += (u64)m * fp.p[j] + t[j];
      t[j - 1] = (u32)c;
      c >>= 32;
    }
    c += t[N];
    t[N - 1] = (u32)c;
    t[N] = t[N + 1] + (u32)(c >> 32);
  }
  // t < 2p here; one conditional subtraction.
  Fp<N> d;
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u64 x = (u64)t[i] - fp.p[i] - borrow;
    d.w[i] = (u32)x;
    borrow = (x >> 63) & 1;
  }
  const bool use_d = (t[N] != 0) | (borrow == 0);
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = use_d ? d.w[i] : t[i];
  return r;
}

template <int N>
__device__ __forceinline__ Fp<N> fp_load_const(const u32* __restrict__ w) {
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = w[i];
  return r;
}

template <int N>
__device__ __forceinline__ Fp<N> fp_to_mont(const Fp<N>& a, const FieldParams& fp) {
  Fp<N> r2;
#pragma unroll
  for (int i = 0; i < N; ++i) r2.w[i] = fp.r2[i];
  return fp_mul<N>(a, r2, fp);
}

template <int N>
__device__ __forceinline__ Fp<N> fp_from_mont(const Fp<N>& a, const FieldParams& fp) {
  Fp<N> one;
#pragma unroll
  for (int i = 0; i < N; ++i) one.w[i] = (i == 0) ? 1u : 0u;
  return fp_mul<N>(a, one, fp);
}

}  // namespace zkgpu
