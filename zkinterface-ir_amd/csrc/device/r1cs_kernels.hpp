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

// ASSIGN = false: compare and record the first failing row per lane.
// ASSIGN = true : C must be a single term with coefficient 1; its slot receives <a,w>*<b,w>.
template <int N, bool ASSIGN>
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
  const uint4* __restrict__ T = args.table + (size_t)lb * args.n_slots * Layout<N>::kRecord + lane;
  const bool lazy = fp.lazy_dot3 != 0 && !(flags & kR1csBIsOne);   // both sums feed the product below
  Fp<N> prod = r1cs_lincomb<N>(args, T, d.first, na, fp, lazy);
  if (!(flags & kR1csBIsOne)) {
    const Fp<N> b = r1cs_lincomb<N>(args, T, d.first + na, nb, fp, lazy);
    prod = fp_mul<N>(prod, b, fp);
  }
  if (ASSIGN) {
    const R1csTerm out = r1cs_load_term(args.terms, d.first + na + nb);
    uint4* __restrict__ O = args.table_out + (size_t)lb * args.n_slots * Layout<N>::kRecord + lane;
    wire_store<N>(O + (size_t)out.slot * Layout<N>::kRecord, prod);
  } else {
    const Fp<N> c = r1cs_lincomb<N>(args, T, d.first + na + nb, nc, fp);
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
