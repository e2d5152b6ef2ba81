// R1CS row kernel: <a,w> * <b,w> against <c,w> for every row and every witness lane.
// Stands in for the zkinterface `Simulator` the reference's tests call on ToR1CSConverter output
// (rust/src/consumers/to_r1cs.rs:583-589,628-634; the crate itself is not under /root/reference).
//
// One wave = one row x 64 witnesses.  Row descriptors and terms are wave-uniform (scalar loads);
// each term is one 2-KiB gather from the wire table (same layout as the replay kernels); the linear
// combinations live in registers.  Algorithmic bytes: 32 B per term and witness.
#pragma once
#include "replay_kernels.hpp"

namespace zkgpu {

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
template <int N>
__device__ __forceinline__ Fp<N> r1cs_term_coef(const R1csArgs& args, const R1csTerm term, const FieldParams& fp) {
  if (term.coef == 0xFFFFFFFFu) {
    Fp<N> c;
#pragma unroll
    for (int i = 0; i < N; ++i) c.w[i] = fp.one[i];
    return c;
  }
  return fp_load_const<N>(args.coefs + (size_t)term.coef * N);
}

// A linear combination, three terms at a time: the three gathers are issued together and, when a
// coefficient other than 1 is present, the three products share one Montgomery reduction (fp_dot).
template <int N>
__device__ __forceinline__ Fp<N> r1cs_lincomb(const R1csArgs& args, const uint4* __restrict__ T, u32 t0, u32 n,
                                              const FieldParams& fp) {
  Fp<N> acc;
#pragma unroll
  for (int i = 0; i < N; ++i) acc.w[i] = 0;
  u32 t = t0;
  const u32 end = t0 + n;
  while (t < end) {
    const u32 m = min(3u, end - t);
    const R1csTerm e0 = args.terms[t];
    const R1csTerm e1 = args.terms[t + (m > 1 ? 1 : 0)];
    const R1csTerm e2 = args.terms[t + (m > 2 ? 2 : 0)];
    const bool plain = e0.coef == 0xFFFFFFFFu && (m < 2 || e1.coef == 0xFFFFFFFFu) && (m < 3 || e2.coef == 0xFFFFFFFFu);
    Fp<N> v[3];
    v[0] = r1cs_term_value<N>(e0, T, fp);
    if (m > 1) v[1] = r1cs_term_value<N>(e1, T, fp);
    if (m > 2) v[2] = r1cs_term_value<N>(e2, T, fp);
    if (plain) {
      acc = fp_add<N>(acc, v[0], fp);
      if (m > 1) acc = fp_add<N>(acc, v[1], fp);
      if (m > 2) acc = fp_add<N>(acc, v[2], fp);
    } else if (m == 3) {
      Fp<N> c[3] = {r1cs_term_coef<N>(args, e0, fp), r1cs_term_coef<N>(args, e1, fp), r1cs_term_coef<N>(args, e2, fp)};
      acc = fp_add<N>(acc, fp_dot<N, 3>(v, c, fp), fp);
    } else if (m == 2) {
      Fp<N> v2[2] = {v[0], v[1]};
      Fp<N> c2[2] = {r1cs_term_coef<N>(args, e0, fp), r1cs_term_coef<N>(args, e1, fp)};
      acc = fp_add<N>(acc, fp_dot<N, 2>(v2, c2, fp), fp);
    } else {
      acc = fp_add<N>(acc, fp_mul<N>(v[0], r1cs_term_coef<N>(args, e0, fp), fp), fp);
    }
    t += m;
  }
  return acc;
}

// ASSIGN = false: compare and record the first failing row per lane.
// ASSIGN = true : C must be a single term with coefficient 1; its slot receives <a,w>*<b,w>.
template <int N, bool ASSIGN>
__global__ __launch_bounds__(256) void r1cs_row_kernel(const R1csArgs args, const FieldParams fp) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 lb = blockIdx.y;
  const u32 r = blockIdx.x * (blockDim.x >> 6) + wave;
  if (r >= args.n_rows) return;
  const u32 row = args.first_row + r;
  const R1csRow d = args.rows[row];
  const u32 na = d.counts & 0xFF, nb = (d.counts >> 8) & 0xFF, nc = (d.counts >> 16) & 0xFF, flags = d.counts >> 24;
  const uint4* __restrict__ T = args.table + (size_t)lb * args.n_slots * Layout<N>::kRecord + lane;
  Fp<N> prod = r1cs_lincomb<N>(args, T, d.first, na, fp);
  if (!(flags & kR1csBIsOne)) {
    const Fp<N> b = r1cs_lincomb<N>(args, T, d.first + na, nb, fp);
    prod = fp_mul<N>(prod, b, fp);
  }
  if (ASSIGN) {
    const R1csTerm out = args.terms[d.first + na + nb];
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

}  // namespace zkgpu
