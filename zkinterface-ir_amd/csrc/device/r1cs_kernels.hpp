// R1CS row kernel: <a,w> * <b,w> against <c,w> for every row and every witness lane.
// Stands in for the zkinterface `Simulator` the reference's tests call on ToR1CSConverter output
// (rust/src/consumers/to_r1cs.rs:583-589,628-634; the crate itself is not under /root/reference).
//
// One wave = one row x 64 witnesses.  Row descriptors and terms are wave-uniform (scalar loads);
// each term is one 2-KiB gather from the wire table (same layout as the replay kernels); the linear
// combinations live in registers.  Algorithmic bytes: 32 B per term and witness.
#pragma once
#include "replay_kernels.hpp"

#ifndef ZKGPU_R1CS_WAVES
#define ZKGPU_R1CS_WAVES 1   // minimum waves per SIMD asked of the register allocator (1 = no constraint)
#endif

namespace zkgpu {

// wave-uniform row / term descriptors on the scalar path
__device__ __forceinline__ R1csTerm r1cs_load_term(const R1csTerm* terms, u32 t) {
  typedef const u32 __attribute__((address_space(4))) cu32;
  cu32* q = (cu32*)(unsigned long long)(terms + __builtin_amdgcn_readfirstlane(t));
  R1csTerm e;
  e.slot = q[0];
  e.coef = q[1];
  return e;
}

template <int N>
__device__ __forceinline__ Fp<N> r1cs_term_value(const R1csTerm term, const uint4* __restrict__ T,
                                                 const FieldParams& fp) {
  Fp<N> v;
  if (term.slot == 0xFFFFFFFFu) {
#pragma unroll
    for (int i = 0; i < N; ++i) v.w[i] = fp.one[i];
  } else {
    v = wire_load<N>(T + (size_t)term.slot * Layout<N>::kRecord);
  }
  return v;
}

// A linear combination, three terms at a time: the three gathers are issued together and, when a coefficient other
// than 1 is present, the three products share one Montgomery reduction (fp_dot) with the coefficients as scalar
// operands.  The first chunk IS the accumulator (no add to zero); every chunk value is canonical.
// lazy: the caller multiplies the result by another such sum (see FieldParams::lazy_dot3): a combination that is one
// chunk of three products then skips its conditional subtraction.
template <int N>
__device__ __forceinline__ Fp<N> r1cs_lincomb(const R1csArgs& args, const uint4* __restrict__ T, u32 t0, u32 n,
                                              const FieldParams& fp, bool lazy = false) {
  Fp<N> acc;
  if (n == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) acc.w[i] = 0;
    return acc;
  }
  const u32 end = t0 + n;
  for (u32 t = t0; t < end;) {
    const u32 m = min(3u, end - t);
    const R1csTerm e0 = r1cs_load_term(args.terms, t);
    const R1csTerm e1 = r1cs_load_term(args.terms, t + (m > 1 ? 1 : 0));
    const R1csTerm e2 = r1cs_load_term(args.terms, t + (m > 2 ? 2 : 0));
    const bool plain = e0.coef == 0xFFFFFFFFu && (m < 2 || e1.coef == 0xFFFFFFFFu) && (m < 3 || e2.coef == 0xFFFFFFFFu);
    Fp<N> part;
    if (plain) {
      part = r1cs_term_value<N>(e0, T, fp);
      if (m > 1) {
        const Fp<N> v1 = r1cs_term_value<N>(e1, T, fp);
        if (m > 2) {
          const Fp<N> v2 = r1cs_term_value<N>(e2, T, fp);
          part = fp_add<N>(fp_add<N>(part, v1, fp), v2, fp);
        } else {
          part = fp_add<N>(part, v1, fp);
        }
      }
    } else {
      auto coef = [&](const R1csTerm& e) { return fp_load_uniform<N>(args.coefs, e.coef == 0xFFFFFFFFu ? args.one_coef : e.coef); };
      if (m == 3) {
        const Fp<N> v[3] = {r1cs_term_value<N>(e0, T, fp), r1cs_term_value<N>(e1, T, fp), r1cs_term_value<N>(e2, T, fp)};
        const FpS<N> c[3] = {coef(e0), coef(e1), coef(e2)};
        part = fp_dot<N, 3>(v, c, fp, (lazy && n == 3) ? 0u : fp.dot_rounds[2]);
      } else if (m == 2) {
        const Fp<N> v[2] = {r1cs_term_value<N>(e0, T, fp), r1cs_term_value<N>(e1, T, fp)};
        const FpS<N> c[2] = {coef(e0), coef(e1)};
        part = fp_dot<N, 2>(v, c, fp, fp.dot_rounds[1]);
      } else {
        const Fp<N> v[1] = {r1cs_term_value<N>(e0, T, fp)};
        const FpS<N> c[1] = {coef(e0)};
        part = fp_dot<N, 1>(v, c, fp, fp.dot_rounds[0]);
      }
    }
    acc = t == t0 ? part : fp_add<N>(acc, part, fp);
    t += m;
  }
  return acc;
}

// ---- coefficient classes (args.hpp kR1csClass*) -----------------------------------------------------------------
// p - a as an integer (p for a = 0: the callers below add or multiply it, they never compare it)
template <int N>
__device__ __forceinline__ Fp<N> r1cs_p_minus(const Fp<N>& a, const FieldParams& fp) {
  u32 pv[N];
  p_words<N>(pv, fp);
  Fp<N> r;
  u64 bw;
  carry_chain<N, true>(r.w, pv, a.w, bw);
  return r;
}
// -a mod p, canonical
template <int N>
__device__ __forceinline__ Fp<N> r1cs_neg(const Fp<N>& a, const FieldParams& fp) {
  const Fp<N> d = r1cs_p_minus<N>(a, fp);
  const bool z = fp_is_zero<N>(a);
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = z ? 0u : d.w[i];
  return r;
}

// class unit: every coefficient is +1 or -1 -- a chain of additions, the negative terms negated first
template <int N>
__device__ __forceinline__ Fp<N> r1cs_lincomb_unit(const R1csArgs& args, const uint4* __restrict__ T, u32 t0, u32 n, const FieldParams& fp) {
  Fp<N> acc;
  const u32 end = t0 + n;
  for (u32 t = t0; t < end;) {
    const u32 m = min(3u, end - t);
    const R1csTerm e0 = r1cs_load_term(args.terms, t);
    const R1csTerm e1 = r1cs_load_term(args.terms, t + (m > 1 ? 1 : 0));
    const R1csTerm e2 = r1cs_load_term(args.terms, t + (m > 2 ? 2 : 0));
    Fp<N> v0 = r1cs_term_value<N>(e0, T, fp), v1 = r1cs_term_value<N>(e1, T, fp), v2 = r1cs_term_value<N>(e2, T, fp);   // the three gathers together
    if (e0.coef >> 31) v0 = r1cs_neg<N>(v0, fp);
    acc = t == t0 ? v0 : fp_add<N>(acc, v0, fp);
    if (m > 1) {
      if (e1.coef >> 31) v1 = r1cs_neg<N>(v1, fp);
      acc = fp_add<N>(acc, v1, fp);
    }
    if (m > 2) {
      if (e2.coef >> 31) v2 = r1cs_neg<N>(v2, fp);
      acc = fp_add<N>(acc, v2, fp);
    }
    t += m;
  }
  return acc;
}

// k Montgomery word rounds on an integer of N + 2 words: x -> (x + (x * n0inv mod 2^32) * p) / 2^32, which is
// x * 2^-32 mod p and below x / 2^32 + p
template <int N>
__device__ __forceinline__ void r1cs_word_rounds(u32 (&x)[N + 2], u32 k, const FieldParams& fp) {
  for (u32 r = 0; r < k; ++r) {
    const u32 q = x[0] * fp.n0inv;
    u64 c = ((u64)q * fp.p[0] + x[0]) >> 32;
#pragma unroll
    for (int i = 1; i < N; ++i) {
      const u64 y = (u64)q * fp.p[i] + x[i] + c;
      x[i - 1] = (u32)y;
      c = y >> 32;
    }
    u64 y = (u64)x[N] + c;
    x[N - 1] = (u32)y;
    y = (u64)x[N + 1] + (y >> 32);
    x[N] = (u32)y;
    x[N + 1] = (u32)(y >> 32);
  }
}
// v * 2^(-32 k) mod p of a canonical v, canonical (each round leaves a value below p + p / 2^32)
template <int N>
__device__ __forceinline__ Fp<N> r1cs_scale_down(const Fp<N>& v, u32 k, const FieldParams& fp) {
  u32 x[N + 2];
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = v.w[i];
  x[N] = x[N + 1] = 0;
  r1cs_word_rounds<N>(x, k, fp);
  return fp_cond_sub<N>(x, __ballot(x[N] != 0), fp, nullptr, nullptr);
}

// class small: sum of c_i * v_i with |c_i| < 2^31 taken as an INTEGER of N + 2 words (a negative term contributes
// |c_i| * (p - v_i); up to 255 terms: below 2^39 * p), N word products per term; two word rounds bring it below 2 p.
// The result is the combination times 2^-64 -- the row kernel keeps count of that factor (r1cs_row_kernel).
constexpr u32 kR1csSmallRounds = 2;
template <int N>
__device__ __forceinline__ void r1cs_small_term(u32 (&x)[N + 2], const Fp<N>& value, u32 coef, const FieldParams& fp) {
  const Fp<N> v = (coef >> 31) ? r1cs_p_minus<N>(value, fp) : value;
  const u32 mag = coef & 0x7FFFFFFFu;
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const u64 y = (u64)v.w[i] * mag + x[i] + c;   // < 2^63 + 2^33
    x[i] = (u32)y;
    c = y >> 32;
  }
  const u64 y = (u64)x[N] + c;
  x[N] = (u32)y;
  x[N + 1] += (u32)(y >> 32);
}
template <int N>
__device__ __forceinline__ Fp<N> r1cs_lincomb_small(const R1csArgs& args, const uint4* __restrict__ T, u32 t0, u32 n, const FieldParams& fp) {
  u32 x[N + 2];
#pragma unroll
  for (int i = 0; i < N + 2; ++i) x[i] = 0;
  const u32 end = t0 + n;
  for (u32 t = t0; t < end;) {
    const u32 m = min(3u, end - t);
    const R1csTerm e0 = r1cs_load_term(args.terms, t);
    const R1csTerm e1 = r1cs_load_term(args.terms, t + (m > 1 ? 1 : 0));
    const R1csTerm e2 = r1cs_load_term(args.terms, t + (m > 2 ? 2 : 0));
    const Fp<N> v0 = r1cs_term_value<N>(e0, T, fp), v1 = r1cs_term_value<N>(e1, T, fp), v2 = r1cs_term_value<N>(e2, T, fp);
    r1cs_small_term<N>(x, v0, e0.coef, fp);
    if (m > 1) r1cs_small_term<N>(x, v1, e1.coef, fp);
    if (m > 2) r1cs_small_term<N>(x, v2, e2.coef, fp);
    t += m;
  }
  r1cs_word_rounds<N>(x, kR1csSmallRounds, fp);
  return fp_cond_sub<N>(x, __ballot(x[N] != 0), fp, nullptr, nullptr);
}

// a combination of any class; `rounds` receives the power of 2^-32 its value carries (0 but for class small)
template <int N, bool CLASSES>
__device__ __forceinline__ Fp<N> r1cs_lincomb_of(u32 cls, const R1csArgs& args, const uint4* __restrict__ T, u32 t0, u32 n,
                                                const FieldParams& fp, bool lazy, u32& rounds) {
  rounds = 0;
  if constexpr (CLASSES) {
    if (cls == kR1csClassUnit) return r1cs_lincomb_unit<N>(args, T, t0, n, fp);
    if (cls == kR1csClassSmall) {
      rounds = kR1csSmallRounds;
      return r1cs_lincomb_small<N>(args, T, t0, n, fp);
    }
  }
  return r1cs_lincomb<N>(args, T, t0, n, fp, lazy);
}

// ASSIGN = false: compare and record the first failing row per lane.
// ASSIGN = true : C must be a single term with coefficient 1; its slot receives <a,w>*<b,w>.
// CLASSES: rows with combinations of class unit / small (args.hpp).  A small-class sum comes scaled by 2^-64; the
// product of the row then carries 2^-64 or 2^-128, and the check compares it with C brought to the same scale by
// word rounds (N + 1 word products each) -- ASSIGN undoes the scale with one product by the Montgomery form of
// 2^64 / 2^128 (behind `one` in the pool).
template <int N, bool ASSIGN, bool CLASSES>
__global__ __launch_bounds__(256, ZKGPU_R1CS_WAVES) void r1cs_row_kernel(const R1csArgs args, const FieldParams fp) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 lb = blockIdx.y;
  const u32 r = blockIdx.x * (blockDim.x >> 6) + wave;
  if (r >= args.n_rows) return;
  const u32 row = args.first_row + r;
  typedef const u32 __attribute__((address_space(4))) cu32;
  cu32* dq = (cu32*)(unsigned long long)(args.rows + __builtin_amdgcn_readfirstlane(row));
  R1csRow d;
  d.first = dq[0];
  d.counts = dq[1];
  const u32 na = d.counts & 0xFF, nb = (d.counts >> 8) & 0xFF, nc = (d.counts >> 16) & 0xFF, flags = d.counts >> 24;
  const u32 cls_a = (flags >> kR1csClassShiftA) & 3, cls_b = (flags >> kR1csClassShiftB) & 3, cls_c = (flags >> kR1csClassShiftC) & 3;
  const uint4* __restrict__ T = args.table + (size_t)lb * args.n_slots * Layout<N>::kRecord + lane;
  // both sums feed the product below (and both are sums of Montgomery products: class full)
  const bool lazy = fp.lazy_dot3 != 0 && !(flags & kR1csBIsOne) && (!CLASSES || (cls_a == kR1csClassFull && cls_b == kR1csClassFull));
  u32 scale, rounds_b = 0;
  Fp<N> prod = r1cs_lincomb_of<N, CLASSES>(cls_a, args, T, d.first, na, fp, lazy, scale);
  if (!(flags & kR1csBIsOne)) {
    const Fp<N> b = r1cs_lincomb_of<N, CLASSES>(cls_b, args, T, d.first + na, nb, fp, lazy, rounds_b);
    prod = fp_mul<N>(prod, b, fp);
  }
  scale += rounds_b;   // prod = <a,w> * <b,w> * 2^(-32 scale)
  if (ASSIGN) {
    if constexpr (CLASSES) {
      if (scale) {
        const FpS<N> fix = fp_load_uniform<N>(args.coefs, args.one_coef + scale / kR1csSmallRounds);
        const Fp<N> v[1] = {prod};
        const FpS<N> c[1] = {fix};
        prod = fp_dot<N, 1>(v, c, fp, fp.dot_rounds[0]);
      }
    }
    const R1csTerm out = r1cs_load_term(args.terms, d.first + na + nb);
    uint4* __restrict__ O = args.table_out + (size_t)lb * args.n_slots * Layout<N>::kRecord + lane;
    wire_store<N>(O + (size_t)out.slot * Layout<N>::kRecord, prod);
  } else {
    u32 rounds_c;
    Fp<N> c = r1cs_lincomb_of<N, CLASSES>(cls_c, args, T, d.first + na + nb, nc, fp, false, rounds_c);
    if constexpr (CLASSES) {
      if (scale > rounds_c) c = r1cs_scale_down<N>(c, scale - rounds_c, fp);
      else if (scale < rounds_c) prod = r1cs_scale_down<N>(prod, rounds_c - scale, fp);
    }
    u32 diff = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) diff |= prod.w[i] ^ c.w[i];
    const u32 lane_g = lb * 64 + lane;
    const bool bad = diff != 0 && lane_g < args.batch;
    if (__ballot(bad) != 0ull) {
      if (bad) atomicMin(&args.first_fail[lane_g], row);
    }
  }
}

// ---- quotient ("correction") wires of ToR1CSConverter with use_correction (to_r1cs.rs:163-211, :213-260, :262-359):
// every add / mul / add_constant / mul_constant call gets a second variable q = (a op b) / p, the integer quotient
// that makes `a op b = out + q * p` hold over the integers (for an R1CS over a larger field).  q is not a field
// operation, but it needs no division: a op b - out = q * p exactly and q < 2^(32N), so
//     q = ((a op b) - out) * p^{-1}  mod 2^(32N)
// -- the low halves of two products.  One wave = one call x 64 witnesses; operands come out of the retain_all wire
// table (Montgomery form -> canonical), the quotient goes out as canonical little-endian words.
template <int N>
__device__ __forceinline__ void mul_low(const u32 (&a)[N], const u32 (&b)[N], u32 (&r)[N]) {   // a * b mod 2^(32N)
  u64 acc = 0;
  u32 carry_hi = 0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
#pragma unroll
    for (int i = 0; i <= k; ++i) {
      const u64 prod = (u64)a[i] * b[k - i];
      acc += prod;
      carry_hi += acc < prod;
    }
    r[k] = (u32)acc;
    acc = (acc >> 32) | ((u64)carry_hi << 32);
    carry_hi = 0;
  }
}

template <int N>
__global__ __launch_bounds__(256) void r1cs_correction_kernel(const R1csCorrArgs args, const FieldParams fp) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 lb = blockIdx.y;
  const u32 k = blockIdx.x * (blockDim.x >> 6) + wave;
  const u32 lane_g = lb * 64 + lane;
  if (k >= args.n_calls) return;
  const R1csCorrCall c = args.calls[k];
  const uint4* __restrict__ T = args.table + (size_t)lb * args.n_slots * Layout<N>::kRecord + lane;
  const Fp<N> av = fp_from_mont<N>(wire_load<N>(T + (size_t)c.a * Layout<N>::kRecord), fp);
  const Fp<N> ov = fp_from_mont<N>(wire_load<N>(T + (size_t)c.out * Layout<N>::kRecord), fp);
  u32 a[N], b[N], out[N], low[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { a[i] = av.w[i]; out[i] = ov.w[i]; }
  if (c.flags & kCorrConstB) {   // the raw constant, as the call got it
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] = args.consts[(size_t)c.b * N + i];
  } else {
    const Fp<N> bv = fp_from_mont<N>(wire_load<N>(T + (size_t)c.b * Layout<N>::kRecord), fp);
#pragma unroll
    for (int i = 0; i < N; ++i) b[i] = bv.w[i];
  }
  if (c.flags & kCorrMul) {
    mul_low<N>(a, b, low);
  } else {
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      carry += (u64)a[i] + b[i];
      low[i] = (u32)carry;
      carry >>= 32;
    }
  }
  u64 borrow = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {   // (a op b) - out  mod 2^(32N)
    const u64 d = (u64)low[i] - out[i] - borrow;
    low[i] = (u32)d;
    borrow = (d >> 63) & 1;
  }
  u32 pinv[N], q[N];
#pragma unroll
  for (int i = 0; i < N; ++i) pinv[i] = args.pinv[i];
  mul_low<N>(low, pinv, q);
  if (lane_g < args.batch) {
    u32* o = args.out + ((size_t)lane_g * args.n_calls + k) * N;
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = q[i];
  }
}

}  // namespace zkgpu
