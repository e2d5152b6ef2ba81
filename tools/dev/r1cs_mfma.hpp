// Linear combinations of the R1CS row kernel on the matrix cores (256-bit fields).
//
// A combination sum_t c_t * v_t has WAVE-UNIFORM coefficients and one value per witness lane: as a product of digit
// strings it is a matrix product -- D[n][m] = sum_k C[n][k] * V[k][m] with V[k][m] digit k of witness m's value and
// C[n][k] = digit n - k of the coefficient (a Toeplitz matrix), D[n][m] digit column n of witness m's double-width
// product.  v_mfma_i32_32x32x32_i8 takes signed bytes, so both factors are recoded into SIGNED base-256 digits
// (x + 0x80..80, every byte ^ 0x80: digit = byte - 128, no carry out while x + 0x8080..80 < 2^256 -- the host checks
// that p allows it, R1csArgs::coef_strings is null otherwise); a term is one K step of 32 digits; the 63 digit columns
// are two tiles of 32 rows, the 64 witnesses two tiles of 32 columns: 4 MFMAs per term.  With the values as the B
// operand the result comes out with the WITNESS on the lane (C/D layout: column = lane & 31), so what is left is lane
// local: 64 column sums of at most 3 * 32 * 2^14 are folded into 16 words and Montgomery-reduced as the sum of
// products of fp_dot is.  The word products of the three N x N multiplications (3 * 64 of the 264 of fp_dot<N, 3>) move
// to the matrix pipe; the reduction (N^2 + N) stays.
//
// Lane maps (gfx950; checked with exact integers by tools/mfma_dot_probe.hip): A / B fragment = 16 bytes per lane, lane
// l = (r = l & 31, h = l >> 5) holds k = 16 h + j (byte j) of row / column r -- whatever the hardware's numbering of k,
// byte j of lane half h of A meets byte j of lane half h of B.  C/D: column = l & 31, row = (g & 3) + 8 (g >> 2) + 4 h for
// register g.  A lane holds its own witness: v_permlane32_swap trades the upper half of the lower 32 witnesses' digits
// for the lower half of the upper 32 (operands) and the two witness tiles' accumulators (results).
#pragma once
#include "fp_mont.hpp"

namespace zkgpu {

typedef int mfma_v4i __attribute__((ext_vector_type(4)));
typedef int mfma_v16i __attribute__((ext_vector_type(16)));

constexpr u32 kCoefStringWords = 24;   // reversed signed digits of a coefficient at bytes 32..63 of 96, zeros around them

// words of a value -> its 32 signed base-256 digits, four to a word
__device__ __forceinline__ void mfma_recode(u32 (&w)[8]) {
  u64 c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const u64 y = (u64)w[i] + 0x80808080u + c;
    w[i] = (u32)y ^ 0x80808080u;
    c = y >> 32;
  }
}

// lanes 32..63 of a <-> lanes 0..31 of b
__device__ __forceinline__ void mfma_swap32(u32& a, u32& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

// per-lane constants of the coefficient fragments: byte address of dword q of the string (for ds_bpermute) and the byte
// shift, for digit-column tile nt: the 16 bytes start at x0 = 63 - (32 nt + r) + 16 h of the string
struct MfmaLane {
  u32 addr[2], shift[2];
};
__device__ __forceinline__ MfmaLane mfma_lane_constants(u32 lane) {
  MfmaLane c;
  const u32 r = lane & 31, h = lane >> 5;
#pragma unroll
  for (u32 nt = 0; nt < 2; ++nt) {
    const u32 x0 = 63 - (32 * nt + r) + 16 * h;
    c.addr[nt] = (x0 >> 2) * 4;
    c.shift[nt] = x0 & 3;
  }
  return c;
}
// the A fragment of one term for tile nt out of `cs` (lane i holds dword i of the term's coefficient string)
__device__ __forceinline__ mfma_v4i mfma_coef_fragment(u32 cs, const MfmaLane& lc, u32 nt) {
  u32 p[5];
#pragma unroll
  for (int d = 0; d < 5; ++d) p[d] = (u32)__builtin_amdgcn_ds_bpermute((int)(lc.addr[nt] + 4 * d), (int)cs);
  mfma_v4i a;
#pragma unroll
  for (int d = 0; d < 4; ++d) a[d] = (int)__builtin_amdgcn_alignbyte(p[d + 1], p[d], lc.shift[nt]);
  return a;
}

// (hi : lo) = lo + x, the carry word DEFINED (the first accumulation of a column, as ZKGPU_MT0 is)
__device__ __forceinline__ void addw_first(u64& lo, u32& hi, u32 x) {
  u64 s;
  asm("v_mad_u64_u32 %0, %2, %3, 1, %0\n\tv_addc_co_u32_e64 %1, %2, 0, 0, %2" : "+v"(lo), "=&v"(hi), "=&s"(s) : "v"(x));
}

// Montgomery reduction of a 2N-word integer X < K p^2 given word by word (x(k) = word k): X / R mod p, `rounds` conditional
// subtractions (fp.dot_rounds[K - 1] makes it canonical) -- the reduction half of fp_dot
template <int N, int k, class WordOf>
__device__ __forceinline__ void mont_reduce_wide_columns(u64& lo, u32& hi, u32 (&m)[N], u32 (&t)[N + 1], const FieldParams& fp, WordOf&& x) {
  if constexpr (k < 2 * N) {
    addw_first(lo, hi, x(std::integral_constant<int, k>()));
    mont_reduce_column<N, k>(lo, hi, m, t, fp);
    mont_reduce_wide_columns<N, k + 1>(lo, hi, m, t, fp, x);
  }
}
template <int N, class WordOf>
__device__ __forceinline__ Fp<N> mont_reduce_wide(const FieldParams& fp, u32 rounds, WordOf&& x) {
  u64 lo = 0;
  u32 hi = 0;
  u32 m[N], t[N + 1];
  mont_reduce_wide_columns<N, 0>(lo, hi, m, t, fp, x);
  t[N] = (u32)lo;
  for (u32 round = 0; round < rounds; ++round) {
    u64 bw;
    const u64 top = __ballot(t[N] != 0);
    const Fp<N> y = fp_cond_sub<N>(t, top, fp, &bw);
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = y.w[i];
    u64 dec = top & bw;
    asm("v_subbrev_co_u32_e64 %0, %1, 0, %0, %1" : "+v"(t[N]), "+s"(dec));
  }
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = t[i];
  return r;
}

// sum_{t < n_terms} c_t * v_t * R^-1 mod p (n_terms <= 3, wave-uniform): v[t] the lanes' values, cs[t] the lanes' dwords of
// the coefficient strings.  The same value as fp_dot<8, K>(v, c, fp, rounds).
__device__ __forceinline__ Fp<8> mfma_dot3(Fp<8> (&v)[3], const u32 (&cs)[3], u32 n_terms, const MfmaLane& lc, const FieldParams& fp,
                                           u32 rounds) {
  mfma_v16i acc[2][2];   // [digit-column tile][witness tile]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[a][b][g] = 0;
#pragma unroll
  for (u32 t = 0; t < 3; ++t) {
    if (t < n_terms) {
      mfma_recode(v[t].w);
#pragma unroll
      for (int i = 0; i < 4; ++i) mfma_swap32(v[t].w[i], v[t].w[4 + i]);
      mfma_v4i b0, b1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        b0[i] = (int)v[t].w[i];        // witnesses 0..31: digits 16 h ..
        b1[i] = (int)v[t].w[4 + i];    // witnesses 32..63
      }
#pragma unroll
      for (u32 nt = 0; nt < 2; ++nt) {
        const mfma_v4i a = mfma_coef_fragment(cs[t], lc, nt);
        acc[nt][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b0, acc[nt][0], 0, 0, 0);
        acc[nt][1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1, acc[nt][1], 0, 0, 0);
      }
    }
  }
  // every lane its own witness: acc[nt][0][g] = column 32 nt + (g & 3) + 8 (g >> 2), acc[nt][1][g] = that + 4
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      u32 x = (u32)acc[nt][0][g], y = (u32)acc[nt][1][g];
      mfma_swap32(x, y);
      acc[nt][0][g] = (int)x;
      acc[nt][1][g] = (int)y;
    }
  // word w of the double-width integer = columns 4 w .. 4 w + 3 (signed, below 2^21 in magnitude) and the carry so far
  long long carry = 0;
  u32 X[16];
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const int nt = w >> 3, q = w & 7, g0 = 4 * (q >> 1);
    const mfma_v16i& A = acc[nt][q & 1];
    const int u = A[g0] + A[g0 + 1] * 256, s = A[g0 + 2] + A[g0 + 3] * 256;    // below 2^30 in magnitude
    const long long y = carry + (long long)u + (long long)s * 65536;
    X[w] = (u32)y;
    carry = y >> 32;
  }
  return mont_reduce_wide<8>(fp, rounds, [&](auto k) { return X[decltype(k)::value]; });
}

}  // namespace zkgpu
