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
// Requirements: p odd, p < 2^(32*N), N <= 16 (512 bits).  p = 2 is handled by the Boolean path.
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
template <int N>
__device__ __forceinline__ Fp<N> fp_cond_sub(const u32* t, u64 force, const FieldParams& fp, u64* borrow_out, const u32* pv = nullptr);

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

// a >= p ?  (defined behind the carry chains below)
// ---- carry chains ----------------------------------------------------------------------------------------------------
// hipcc lowers `c += (u64)a + b; s = (u32)c; c >>= 32` and its subtracting twin to 64-bit arithmetic: five VALU
// instructions per word (a 32-bit add, a zero-extension move, two 64-bit adds, a shift) where the hardware has one --
// v_addc_co_u32 / v_subb_co_u32 with the carry in an SGPR pair.  A modular add came out as ~105 VALU instructions, the
// conditional subtraction that ends every Montgomery product as ~45.  Here a chain is a few asm statements of up to four
// words each, the carry travelling between them in a 64-bit lane mask (an SGPR pair: "s" operands of VOP3b).
// One SGPR (or VCC) is all the constant bus of gfx9 lets an instruction read, and the carry-in is one: the OTHER operands
// of a chain have to be VGPRs -- the words of p are copied into registers for it (v_mov from the kernarg SGPRs; the
// compiler rematerialises them where it is short of registers).
template <int K, bool FIRST, bool SUB>
__device__ __forceinline__ void carry_chunk(u32* r, const u32* a, const u32* b, u64& c) {
  static_assert(K == 2 || K == 4, "");
  if constexpr (K == 4) {
    if constexpr (FIRST) {
      if constexpr (SUB)
        asm("v_sub_co_u32_e64 %0, %4, %5, %9\n\tv_subb_co_u32_e64 %1, %4, %6, %10, %4\n\tv_subb_co_u32_e64 %2, %4, %7, %11, %4\n\t"
            "v_subb_co_u32_e64 %3, %4, %8, %12, %4"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&s"(c)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
      else
        asm("v_add_co_u32_e64 %0, %4, %5, %9\n\tv_addc_co_u32_e64 %1, %4, %6, %10, %4\n\tv_addc_co_u32_e64 %2, %4, %7, %11, %4\n\t"
            "v_addc_co_u32_e64 %3, %4, %8, %12, %4"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&s"(c)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    } else {
      if constexpr (SUB)
        asm("v_subb_co_u32_e64 %0, %4, %5, %9, %4\n\tv_subb_co_u32_e64 %1, %4, %6, %10, %4\n\tv_subb_co_u32_e64 %2, %4, %7, %11, %4\n\t"
            "v_subb_co_u32_e64 %3, %4, %8, %12, %4"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "+s"(c)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
      else
        asm("v_addc_co_u32_e64 %0, %4, %5, %9, %4\n\tv_addc_co_u32_e64 %1, %4, %6, %10, %4\n\tv_addc_co_u32_e64 %2, %4, %7, %11, %4\n\t"
            "v_addc_co_u32_e64 %3, %4, %8, %12, %4"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "+s"(c)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    }
  } else {
    if constexpr (FIRST) {
      if constexpr (SUB)
        asm("v_sub_co_u32_e64 %0, %2, %3, %5\n\tv_subb_co_u32_e64 %1, %2, %4, %6, %2"
            : "=&v"(r[0]), "=&v"(r[1]), "=&s"(c) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
      else
        asm("v_add_co_u32_e64 %0, %2, %3, %5\n\tv_addc_co_u32_e64 %1, %2, %4, %6, %2"
            : "=&v"(r[0]), "=&v"(r[1]), "=&s"(c) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
    } else {
      if constexpr (SUB)
        asm("v_subb_co_u32_e64 %0, %2, %3, %5, %2\n\tv_subb_co_u32_e64 %1, %2, %4, %6, %2"
            : "=&v"(r[0]), "=&v"(r[1]), "+s"(c) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
      else
        asm("v_addc_co_u32_e64 %0, %2, %3, %5, %2\n\tv_addc_co_u32_e64 %1, %2, %4, %6, %2"
            : "=&v"(r[0]), "=&v"(r[1]), "+s"(c) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
    }
  }
}

template <int N, bool SUB, int I = 0>
__device__ __forceinline__ void carry_chain(u32* r, const u32* a, const u32* b, u64& c) {
  static_assert(N % 2 == 0, "field widths are whole 64-bit limbs");
  if constexpr (I < N) {
    constexpr int K = (N - I >= 4) ? 4 : 2;
    carry_chunk<K, I == 0, SUB>(r + I, a + I, b + I, c);
    carry_chain<N, SUB, I + K>(r, a, b, c);
  }
}

// r[i] = lanes of `mask` ? d[i] : s[i] (the mask is a 64-bit lane mask in an SGPR pair)
template <int N, int I = 0>
__device__ __forceinline__ void select_words(u32* r, const u32* s, const u32* d, u64 mask) {
  if constexpr (I < N) {
    if constexpr (N - I >= 4) {
      asm("v_cndmask_b32_e64 %0, %4, %8, %12\n\tv_cndmask_b32_e64 %1, %5, %9, %12\n\tv_cndmask_b32_e64 %2, %6, %10, %12\n\t"
          "v_cndmask_b32_e64 %3, %7, %11, %12"
          : "=&v"(r[I]), "=&v"(r[I + 1]), "=&v"(r[I + 2]), "=&v"(r[I + 3])
          : "v"(s[I]), "v"(s[I + 1]), "v"(s[I + 2]), "v"(s[I + 3]), "v"(d[I]), "v"(d[I + 1]), "v"(d[I + 2]), "v"(d[I + 3]), "s"(mask));
      select_words<N, I + 4>(r, s, d, mask);
    } else {
      asm("v_cndmask_b32_e64 %0, %2, %4, %6\n\tv_cndmask_b32_e64 %1, %3, %5, %6"
          : "=&v"(r[I]), "=&v"(r[I + 1]) : "v"(s[I]), "v"(s[I + 1]), "v"(d[I]), "v"(d[I + 1]), "s"(mask));
      select_words<N, I + 2>(r, s, d, mask);
    }
  }
}

// the words of p in VGPRs (see above: a carry chain cannot read them from SGPRs)
template <int N>
__device__ __forceinline__ void p_words(u32 (&pv)[N], const FieldParams& fp) {
#pragma unroll
  for (int i = 0; i < N; ++i) pv[i] = fp.p[i];
}

// the same, opaque to hipcc: it copies the words once and keeps them (left to itself it sinks the eight v_mov into every
// branch that subtracts p -- cheap to recompute, so it recomputes them: 19 instead of 8 moves per program entry)
template <int N>
__device__ __forceinline__ void p_words_resident(u32 (&pv)[N], const FieldParams& fp) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(pv[i]) : "s"(fp.p[i]));
}

// t - p where that is not negative, else t; `force`: lanes whose value has a set bit above the N words (t >= 2^(32N) > p).
// For t < 2p the result is canonical.  borrow_out (optional): lanes where t < p as N-word integers.  pv_in (optional): the
// words of p already in VGPRs (p_words) -- a kernel that runs several operations per wave copies them once.
template <int N>
__device__ __forceinline__ Fp<N> fp_cond_sub(const u32* t, u64 force, const FieldParams& fp, u64* borrow_out, const u32* pv_in) {
  u32 pv[N], d[N];
  if (pv_in) {
#pragma unroll
    for (int i = 0; i < N; ++i) pv[i] = pv_in[i];
  } else {
    p_words<N>(pv, fp);
  }
  u64 bw;
  carry_chain<N, true>(d, t, pv, bw);
  if (borrow_out) *borrow_out = bw;
  Fp<N> r;
  select_words<N>(r.w, t, d, force | ~bw);
  return r;
}

// a >= p ?
template <int N>
__device__ __forceinline__ bool fp_geq_p(const Fp<N>& a, const FieldParams& fp) {
  u32 pv[N], d[N];
  p_words<N>(pv, fp);
  u64 bw;
  carry_chain<N, true>(d, a.w, pv, bw);
  u32 ge;
  asm("v_cndmask_b32_e64 %0, 1, 0, %1" : "=v"(ge) : "s"(bw));   // no borrow: a >= p
  return ge != 0;
}

// r = (a + b) mod p for canonical a, b.
template <int N>
__device__ __forceinline__ Fp<N> fp_add(const Fp<N>& a, const Fp<N>& b, const FieldParams& fp, const u32* pv = nullptr) {
  u32 s[N];
  u64 carry;
  carry_chain<N, false>(s, a.w, b.w, carry);
  // s >= p  <=>  carry-out of the add, or no borrow in the subtract
  return fp_cond_sub<N>(s, carry, fp, nullptr, pv);
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
  static_assert(K >= 1 && K <= 16, "");
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
__device__ __forceinline__ Fp<N> fp_mul(const Fp<N>& a, const Fp<N>& b, const FieldParams& fp, const u32* pv = nullptr) {
  u64 lo = 0;
  u32 hi = 0;
  u32 m[N], t[N + 1];
  fp_mul_columns<N, 0>(lo, hi, a, b, m, t, fp);
  t[N] = (u32)lo;
  // t < 2p here; one conditional subtraction.
  return fp_cond_sub<N>(t, __ballot(t[N] != 0), fp, nullptr, pv);
}

// ---- the same product, written for LATENCY -------------------------------------------------------------------------
// fp_mul above accumulates a column's word products into ONE 96-bit accumulator, each v_mad_u64_u32 waiting for the one
// before it and each v_addc_co_u32 for its own mad: fine with eight waves per SIMD (some other wave always has an
// instruction ready), but a STRAND is one workgroup per lane block -- one wave per SIMD at most -- and there the
// dependent chain is what takes the time (tools/mul_latency.hip: about 2.4 times the cycles per product of this form).
// Here every column k of the double-width result has an accumulator of its own (lo64 : hi32), a row of products
// x[j] * y goes to columns that are all different, and a statement issues four mads with four different carry registers
// (SGPR pairs, VOP3b) before the four add-with-carries that consume them: consecutive instructions never depend on each
// other.  Operand-scanning Montgomery: step k clears the low word of column k with m = lo * n0inv, m * p[j] going to
// columns k .. k + N - 1, and hands the column's upper words on to columns k + 1 and k + 2.  Same result as fp_mul, bit
// for bit (canonical out); twice the registers.
__device__ __forceinline__ void madc4(u64& c0, u64& c1, u64& c2, u64& c3, u32& h0, u32& h1, u32& h2, u32& h3, u32 x0, u32 x1, u32 x2,
                                      u32 x3, u32 y) {
  u64 s0, s1, s2, s3;
  asm("v_mad_u64_u32 %0, %8, %12, %16, %0\n\t"
      "v_mad_u64_u32 %1, %9, %13, %16, %1\n\t"
      "v_mad_u64_u32 %2, %10, %14, %16, %2\n\t"
      "v_mad_u64_u32 %3, %11, %15, %16, %3\n\t"
      "v_addc_co_u32_e64 %4, %8, 0, %4, %8\n\t"
      "v_addc_co_u32_e64 %5, %9, 0, %5, %9\n\t"
      "v_addc_co_u32_e64 %6, %10, 0, %6, %10\n\t"
      "v_addc_co_u32_e64 %7, %11, 0, %7, %11"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
      : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(y));
}
// ... with the first factors wave-uniform (the words of p, in SGPRs: one scalar operand per instruction) and the second a VGPR
__device__ __forceinline__ void madc4_s(u64& c0, u64& c1, u64& c2, u64& c3, u32& h0, u32& h1, u32& h2, u32& h3, u32 x0, u32 x1, u32 x2,
                                        u32 x3, u32 y) {
  u64 s0, s1, s2, s3;
  asm("v_mad_u64_u32 %0, %8, %16, %12, %0\n\t"
      "v_mad_u64_u32 %1, %9, %16, %13, %1\n\t"
      "v_mad_u64_u32 %2, %10, %16, %14, %2\n\t"
      "v_mad_u64_u32 %3, %11, %16, %15, %3\n\t"
      "v_addc_co_u32_e64 %4, %8, 0, %4, %8\n\t"
      "v_addc_co_u32_e64 %5, %9, 0, %5, %9\n\t"
      "v_addc_co_u32_e64 %6, %10, 0, %6, %10\n\t"
      "v_addc_co_u32_e64 %7, %11, 0, %7, %11"
      : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
      : "s"(x0), "s"(x1), "s"(x2), "s"(x3), "v"(y));
}
// (hi : lo) += x, a 32-bit word added to a 96-bit accumulator
__device__ __forceinline__ void addw(u64& lo, u32& hi, u32 x) {
  u64 s;
  asm("v_mad_u64_u32 %0, %2, 1, %3, %0\n\tv_addc_co_u32_e64 %1, %2, 0, %1, %2" : "+v"(lo), "+v"(hi), "=&s"(s) : "v"(x));
}

template <int N>
__device__ __forceinline__ Fp<N> fp_mul_wide(const Fp<N>& a, const Fp<N>& b, const FieldParams& fp, const u32* pv = nullptr) {
  static_assert(N % 4 == 0 || N % 4 == 2, "");
  u64 lo[2 * N + 2];
  u32 hi[2 * N + 2];
#pragma unroll
  for (int k = 0; k < 2 * N + 2; ++k) { lo[k] = 0; hi[k] = 0; }
  // the double-width product: row i = a * b[i] into columns i .. i + N - 1
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j + 4 <= N; j += 4)
      madc4(lo[i + j], lo[i + j + 1], lo[i + j + 2], lo[i + j + 3], hi[i + j], hi[i + j + 1], hi[i + j + 2], hi[i + j + 3], a.w[j], a.w[j + 1],
            a.w[j + 2], a.w[j + 3], b.w[i]);
    if constexpr (N % 4 == 2) {
      ZKGPU_MADC(lo[i + N - 2], hi[i + N - 2], a.w[N - 2], b.w[i]);
      ZKGPU_MADC(lo[i + N - 1], hi[i + N - 1], a.w[N - 1], b.w[i]);
    }
  }
  // the reduction, column by column
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const u32 m = (u32)lo[k] * fp.n0inv;
#pragma unroll
    for (int j = 0; j + 4 <= N; j += 4)
      madc4_s(lo[k + j], lo[k + j + 1], lo[k + j + 2], lo[k + j + 3], hi[k + j], hi[k + j + 1], hi[k + j + 2], hi[k + j + 3], fp.p[j],
              fp.p[j + 1], fp.p[j + 2], fp.p[j + 3], m);
    if constexpr (N % 4 == 2) {
      ZKGPU_MADC_S(lo[k + N - 2], hi[k + N - 2], m, fp.p[N - 2]);
      ZKGPU_MADC_S(lo[k + N - 1], hi[k + N - 1], m, fp.p[N - 1]);
    }
    // the low word of column k is zero now: its upper words belong to the next two columns
    addw(lo[k + 1], hi[k + 1], (u32)(lo[k] >> 32));
    addw(lo[k + 2], hi[k + 2], hi[k]);
  }
  u32 t[N + 1];
#pragma unroll
  for (int k = N; k < 2 * N; ++k) {
    t[k - N] = (u32)lo[k];
    addw(lo[k + 1], hi[k + 1], (u32)(lo[k] >> 32));
    addw(lo[k + 2], hi[k + 2], hi[k]);
  }
  t[N] = (u32)lo[2 * N];
  // t < 2p here; one conditional subtraction.
  return fp_cond_sub<N>(t, __ballot(t[N] != 0), fp, nullptr, pv);
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
    u64 bw;
    const u64 top = __ballot(t[N] != 0);
    const Fp<N> x = fp_cond_sub<N>(t, top, fp, &bw);
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = x.w[i];
    // where p was subtracted (top | ~bw) and the N-word subtraction borrowed, the borrow comes out of the top word
    u64 dec = top & bw;
    asm("v_subbrev_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(t[N]), "+s"(dec));
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
