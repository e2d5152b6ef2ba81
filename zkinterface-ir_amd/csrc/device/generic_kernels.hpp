// Any-modulus replay: the fields the Montgomery kernels cannot take.
//
// PlaintextBackend works on BigUint with `% m` after every gate (rust/src/consumers/evaluator.rs:908-938), so it takes
// ANY modulus >= 1 of any width.  The Montgomery kernels (fp_mont.hpp) need an odd modulus of at most 512 bits.  This
// header is the path for the rest -- an even characteristic other than 2 as the only field of a session, one wider than
// 512 bits, and GF(2) where a session changes between GF(2) and another field (bit-packed wires and integers do not
// mix, the reference carries the integers over: evaluator.rs:232-237).  Wires hold CANONICAL residues (no Montgomery
// form) in the same wire table layout (table[lane block][slot][chunk][lane]); products are reduced with Barrett's method
// (Handbook of Applied Cryptography, algorithm 14.42, radix 2^32), which works for every modulus.  Word counts are run
// time values: operands live in thread-private arrays (scratch memory), loops are not unrolled.  It is a correctness
// path: one lane per witness like every other kernel, nothing tuned -- the relations that need it are rare.
//
// The arithmetic is plain C++ on 32-bit words (`__host__ __device__`): the CPU tier runs the same functions against
// Python integers through zkgpu_generic_selftest (capi.cpp).
#pragma once
#include "args.hpp"

namespace zkgpu {

#define ZKGPU_HD __host__ __device__ __forceinline__

// 1 if a >= p, a: len >= k words, p: k words
template <class P>
ZKGPU_HD bool g_geq_p(const u32* a, u32 len, const P* gp) {
  const u32 k = gp->k;
  for (u32 i = len; i-- > k;)
    if (a[i]) return true;
  for (u32 i = k; i-- > 0;)
    if (a[i] != gp->p[i]) return a[i] > gp->p[i];
  return true;
}
template <class P>
ZKGPU_HD void g_sub_p(u32* a, u32 len, const P* gp) {
  const u32 k = gp->k;
  u32 borrow = 0;
  for (u32 i = 0; i < len; ++i) {
    const u64 d = (u64)a[i] - (i < k ? gp->p[i] : 0u) - borrow;
    a[i] = (u32)d;
    borrow = (u32)(d >> 63);
  }
}
ZKGPU_HD bool g_is_zero(const u32* a, u32 n) {
  u32 acc = 0;
  for (u32 i = 0; i < n; ++i) acc |= a[i];
  return acc == 0;
}

// column accumulator of a product scan: 96 bits
struct GAcc {
  u64 lo;
  u32 hi;
  ZKGPU_HD void mac(u32 x, u32 y) {
    const u64 pr = (u64)x * y;
    lo += pr;
    hi += lo < pr ? 1u : 0u;
  }
  ZKGPU_HD u32 shift() {   // the finished column; the rest moves down one word
    const u32 w = (u32)lo;
    lo = (lo >> 32) | ((u64)hi << 32);
    hi = 0;
    return w;
  }
};

// out = x mod p for x < 2^(64 k), x: 2 k words (HAC 14.42: q3 = floor(floor(x / b^(k-1)) * mu / b^(k+1)),
// r = (x - q3 * p) mod b^(k+1), then at most two subtractions of p).  out: n = nwords >= k words.
// p = 2^B (a ring Z / 2^B, e.g. 2^32 or 2^64): x mod p is the low B bits of x -- out[0 .. n) from the low words of x
template <class P>
ZKGPU_HD void g_low_bits(const u32* x, u32 x_words, u32* out, const P* gp) {
  const u32 B = gp->pow2_bits, n = gp->nwords, full = B / 32, rest = B % 32;
  for (u32 i = 0; i < n; ++i) out[i] = i < full && i < x_words ? x[i] : 0u;
  if (rest && full < n && full < x_words) out[full] = x[full] & ((1u << rest) - 1u);
}

template <int CAP, class P>
ZKGPU_HD void g_barrett(const u32* x, u32* out, const P* gp) {
  const u32 k = gp->k, n = gp->nwords;
  if (gp->pow2_bits) {
    g_low_bits(x, 2 * k, out, gp);
    return;
  }
  // q3: columns k + 1 .. 2 k + 2 of q1 * mu, q1 = x[k - 1 .. 2 k - 1] (k + 1 words), mu k + 2 words
  u32 q3[CAP + 2];
  GAcc acc{0, 0};
  for (u32 col = 0; col <= 2 * k + 1; ++col) {
    const u32 i_lo = col > k + 1 ? col - (k + 1) : 0, i_hi = col < k ? col : k;
    for (u32 i = i_lo; i <= i_hi; ++i) acc.mac(x[k - 1 + i], gp->mu[col - i]);
    const u32 w = acc.shift();
    if (col >= k + 1) q3[col - (k + 1)] = w;
  }
  q3[k + 1] = (u32)acc.lo;
  // r = x mod b^(k+1) - (q3 * p) mod b^(k+1), column by column with a running borrow
  u32 r[CAP + 1];
  acc = GAcc{0, 0};
  u32 borrow = 0;
  for (u32 col = 0; col <= k; ++col) {
    const u32 j_hi = col < k ? col : k - 1;
    for (u32 j = 0; j <= j_hi; ++j) acc.mac(q3[col - j], gp->p[j]);
    const u32 m = acc.shift();
    const u64 d = (u64)(col < 2 * k ? x[col] : 0u) - m - borrow;
    r[col] = (u32)d;
    borrow = (u32)(d >> 63);
  }
  for (int round = 0; round < 3 && g_geq_p(r, k + 1, gp); ++round) g_sub_p(r, k + 1, gp);
  for (u32 i = 0; i < n; ++i) out[i] = i < k ? r[i] : 0u;
}

// out = a * b mod p (a, b canonical, n words each)
template <int CAP, class P>
ZKGPU_HD void g_mul(const u32* a, const u32* b, u32* out, const P* gp) {
  const u32 k = gp->k;
  u32 x[2 * CAP];
  GAcc acc{0, 0};
  if (gp->pow2_bits) {   // only the columns below 2^B count: half a product, no reduction
    const u32 kw = (gp->pow2_bits + 31) / 32;
    for (u32 col = 0; col < kw; ++col) {
      for (u32 i = 0; i <= col; ++i) acc.mac(a[i], b[col - i]);
      x[col] = acc.shift();
    }
    g_low_bits(x, kw, out, gp);
    return;
  }
  for (u32 col = 0; col < 2 * k; ++col) {
    const u32 i_lo = col >= k ? col - (k - 1) : 0, i_hi = col < k ? col : k - 1;
    for (u32 i = i_lo; i <= i_hi; ++i) acc.mac(a[i], b[col - i]);
    x[col] = acc.shift();
  }
  g_barrett<CAP>(x, out, gp);
}

// out = a + b mod p
template <int CAP, class P>
ZKGPU_HD void g_add(const u32* a, const u32* b, u32* out, const P* gp) {
  const u32 k = gp->k, n = gp->nwords;
  u32 r[CAP + 1];
  u64 c = 0;
  if (gp->pow2_bits) {   // the sum wraps
    const u32 kw = (gp->pow2_bits + 31) / 32;
    for (u32 i = 0; i < kw; ++i) {
      c += (u64)a[i] + b[i];
      r[i] = (u32)c;
      c >>= 32;
    }
    g_low_bits(r, kw, out, gp);
    return;
  }
  for (u32 i = 0; i < k; ++i) {
    c += (u64)a[i] + b[i];
    r[i] = (u32)c;
    c >>= 32;
  }
  r[k] = (u32)c;
  if (g_geq_p(r, k + 1, gp)) g_sub_p(r, k + 1, gp);
  for (u32 i = 0; i < n; ++i) out[i] = i < k ? r[i] : 0u;
}

// out = raw mod p for a raw value of n words (n <= k + 1 <= 2 k)
template <int CAP, class P>
ZKGPU_HD void g_reduce(const u32* raw, u32* out, const P* gp) {
  const u32 k = gp->k, n = gp->nwords;
  u32 x[2 * CAP];
  for (u32 i = 0; i < 2 * k; ++i) x[i] = i < n ? raw[i] : 0u;
  g_barrett<CAP>(x, out, gp);
}

// (a & b) % p and (a ^ b) % p on canonical values (evaluator.rs:924-933): the conjunction is below both operands, the
// exclusive or below 2^bits(p) <= 2 p
template <class P>
ZKGPU_HD void g_and(const u32* a, const u32* b, u32* out, const P* gp) {
  for (u32 i = 0; i < gp->nwords; ++i) out[i] = a[i] & b[i];
}
template <class P>
ZKGPU_HD void g_xor(const u32* a, const u32* b, u32* out, const P* gp) {
  const u32 n = gp->nwords;
  for (u32 i = 0; i < n; ++i) out[i] = a[i] ^ b[i];
  if (g_geq_p(out, n, gp)) g_sub_p(out, n, gp);
}
template <class P>
ZKGPU_HD void g_indicator(bool one, u32* out, const P* gp) {
  for (u32 i = 0; i < gp->nwords; ++i) out[i] = 0;
  out[0] = one ? 1u : 0u;
}

#ifdef __HIPCC__
// ---- wire table access (the layout of replay_kernels.hpp with a run time chunk count) ----
__device__ __forceinline__ void g_wire_load(const uint4* __restrict__ rec, u32 n, u32* out) {
  for (u32 c = 0; 4 * c < n; ++c) {
    const uint4 v = rec[c * 64];   // (the last chunk of a value of 4 c + 2 words is half used: its upper half holds zeros)
    out[4 * c] = v.x;
    out[4 * c + 1] = v.y;
    if (4 * c + 2 < n) {
      out[4 * c + 2] = v.z;
      out[4 * c + 3] = v.w;
    }
  }
}
__device__ __forceinline__ void g_wire_store(uint4* __restrict__ rec, u32 n, const u32* r) {
  for (u32 c = 0; 4 * c < n; ++c) {
    const bool full = 4 * c + 2 < n;
    rec[c * 64] = make_uint4(r[4 * c], r[4 * c + 1], full ? r[4 * c + 2] : 0u, full ? r[4 * c + 3] : 0u);
  }
}

// the raw value of input stream 0 (instance), 1 (witness) or 2 (carried over) at `position`: its low n words; returns
// whether it has bits above them (replay_kernels.hpp input_load)
__device__ __forceinline__ bool g_stream_load(u32 stream, u32 position, const ReplayArgs& args, u32 lane_g, bool valid, u32 n,
                                              u32* out) {
  typedef const InputAux __attribute__((address_space(4))) AuxS;
  AuxS* aux = (AuxS*)(unsigned long long)args.aux;
  for (u32 i = 0; i < n; ++i) out[i] = 0;
  if (stream == 3) {   // a constant kept as the integer it is (replay_kernels.hpp stream_load)
    const u32* c = args.consts + (size_t)(aux->raw_const_base + position) * n;
    for (u32 i = 0; i < n; ++i) out[i] = c[i];
    return false;
  }
  if (!valid) return false;
  const u32* base;
  u32 n_vals, stride;
  if (stream == 2) {
    base = aux->carry;
    n_vals = aux->n_carry;
    stride = aux->carry_words;
  } else {
    base = reinterpret_cast<const u32*>(stream ? args.wit : args.inst);
    n_vals = stream ? args.n_wit : args.n_inst;
    stride = aux->in_stride_words;
  }
  const u32* p = base + ((size_t)lane_g * n_vals + position) * stride;
  u32 hi = 0;
  for (u32 i = 0; i < stride; ++i) {
    if (i < n) out[i] = p[i];
    else hi |= p[i];
  }
  return hi != 0;
}

template <int CAP, class P>
__device__ __forceinline__ bool g_unreduced_source_is_nonzero(u32 code, const ReplayArgs& args, u32 lane_g, bool lane_valid,
                                                              const P* gp) {
  if (code < 2) return code == 1;
  const u32 q = code - 2;
  u32 raw[CAP];
  const bool too_wide = g_stream_load(q & 3, q >> 2, args, lane_g, lane_valid, gp->nwords, raw);
  return lane_valid && (too_wide || g_geq_p(raw, gp->nwords, gp));
}

// The parameters of a SMALL characteristic (up to eight words: the rings Z / 2^32 .. Z / 2^128, even moduli up to 256 bits) with the
// word counts known at compile time and the words themselves in registers: every loop of the arithmetic above has a
// constant trip count then, its private arrays become registers, and p and mu are not fetched word by word from memory --
// the C2 relation over Z / 2^64 ran in 103 ms on the general kernel (profiles/r04_tuning_sweeps.txt section 5).
template <int KC>
struct SmallParams {
  static constexpr u32 k = KC, nwords = 2 * ((KC + 1) / 2);
  u32 p[KC];
  u32 mu[KC + 2];
  u32 pow2_bits;
};

// One wave = 64 witnesses x `ops_per_wave` consecutive entries of the unfused program (the TapeOp entries of
// replay_kernel; the scheduler makes no fused or pair entries for these fields).
template <int CAP, class P>
__device__ __forceinline__ void replay_generic_body(const ReplayArgs& args, const P* __restrict__ gp) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 chunk = args.xcd_chunks ? (blockIdx.x >> 3) % args.xcd_chunks : blockIdx.x;
  const u32 lb_rel = args.xcd_chunks ? ((blockIdx.x >> 3) / args.xcd_chunks) * 8 + (blockIdx.x & 7) : blockIdx.y;
  const u32 lb = args.lb_base + lb_rel;
  const u32 gw = chunk * (blockDim.x >> 6) + wave;
  const u32 begin = gw * args.ops_per_wave;
  if (begin >= args.n_ops) return;
  const u32 end = min(args.n_ops, begin + args.ops_per_wave);
  const u32 lane_g = lb * 64 + lane;
  const bool lane_valid = lane_g < args.batch;
  const u32 n = gp->nwords;
  const u32 rec = ((n + 3) / 4) * 64;
  uint4* __restrict__ T = args.table + (size_t)lb * args.n_slots * rec + lane;
  typedef const InputAux __attribute__((address_space(4))) AuxS;
  AuxS* aux = (AuxS*)(unsigned long long)args.aux;

  for (u32 i = begin; i < end; ++i) {
    const TapeOp op = args.ops[i];
    u32 a[CAP], b[CAP], r[CAP];
    bool has_out = true;
    switch (op.kind) {
      case OP_ADD:
      case OP_MUL:
        g_wire_load(T + (size_t)op.a * rec, n, a);
        g_wire_load(T + (size_t)op.b * rec, n, b);
        if (op.kind == OP_ADD) g_add<CAP>(a, b, r, gp);
        else g_mul<CAP>(a, b, r, gp);
        break;
      case OP_AND:
      case OP_XOR: {
        // an operand may name the RAW value of an input the relation has only copied (args.hpp kOperandIsSource):
        // PlaintextBackend works on those bits (evaluator.rs:924-933); the result is reduced like any raw value
        bool wide = false;
        if (op.a & kOperandIsSource) wide |= g_stream_load(((op.a & ~kOperandIsSource) - 2) & 3, ((op.a & ~kOperandIsSource) - 2) >> 2, args, lane_g, lane_valid, n, a);
        else g_wire_load(T + (size_t)op.a * rec, n, a);
        if (op.b & kOperandIsSource) wide |= g_stream_load(((op.b & ~kOperandIsSource) - 2) & 3, ((op.b & ~kOperandIsSource) - 2) >> 2, args, lane_g, lane_valid, n, b);
        else g_wire_load(T + (size_t)op.b * rec, n, b);
        if (lane_valid && wide) atomicOr(&args.lane_flags[lane_g], kLaneFlagNonCanonical);
        for (u32 w = 0; w < n; ++w) a[w] = op.kind == OP_AND ? (a[w] & b[w]) : (a[w] ^ b[w]);
        g_reduce<CAP>(a, r, gp);
        break;
      }
      case OP_ADDC:
      case OP_MULC:
        g_wire_load(T + (size_t)op.a * rec, n, a);
        for (u32 w = 0; w < n; ++w) b[w] = args.consts[(size_t)op.b * n + w];
        if (op.kind == OP_ADDC) g_add<CAP>(a, b, r, gp);
        else g_mul<CAP>(a, b, r, gp);
        break;
      case OP_COPY: g_wire_load(T + (size_t)op.a * rec, n, r); break;
      case OP_NZ:
        g_wire_load(T + (size_t)op.a * rec, n, a);
        g_indicator(!g_is_zero(a, n), r, gp);
        break;
      case OP_NOT:   // op.b: the unreduced source behind the operand, if any
        g_wire_load(T + (size_t)op.a * rec, n, a);
        g_indicator(g_is_zero(a, n) && !g_unreduced_source_is_nonzero<CAP>(op.b, args, lane_g, lane_valid, gp), r, gp);
        break;
      case OP_CONST:
        for (u32 w = 0; w < n; ++w) r[w] = args.consts[(size_t)op.a * n + w];
        break;
      case OP_INSTANCE:
      case OP_WITNESS:
      case OP_CARRY: {
        const u32 stream = op.kind == OP_INSTANCE ? 0u : op.kind == OP_WITNESS ? 1u : 2u;
        const bool too_wide = g_stream_load(stream, op.a, args, lane_g, lane_valid, n, a);
        const uint8_t* modes = stream == 0 ? aux->strict_inst : stream == 1 ? aux->strict_wit : aux->strict_carry;
        const u32 mode = modes[op.a];
        // (replay_kernels.hpp input_op: flagged only where the value's bits matter and the limbs cannot hold them, or where it
        // is >= p and the residue will not do; anywhere else a value of any width is reduced, as the reference's arithmetic does)
        if (lane_valid && ((too_wide && (mode == 0xFF || mode == 0x03)) || (mode == 0xFF && g_geq_p(a, n, gp))))
          atomicOr(&args.lane_flags[lane_g], kLaneFlagNonCanonical);
        if (!(too_wide && lane_valid)) {
          g_reduce<CAP>(a, r, gp);
        } else {
          // Horner over the groups of n words: W = 2^(32 n) mod p = (2^(32 n) - 1) mod p + 1
          const u32* base = stream == 2 ? aux->carry : reinterpret_cast<const u32*>(stream ? args.wit : args.inst);
          const u32 n_vals = stream == 2 ? aux->n_carry : (stream ? args.n_wit : args.n_inst);
          const u32 stride = stream == 2 ? aux->carry_words : aux->in_stride_words;
          const u32* q = base + ((size_t)lane_g * n_vals + op.a) * stride;
          u32 W[CAP];   // (one more private array: the groups go through a / b)
          for (u32 w = 0; w < n; ++w) a[w] = 0xFFFFFFFFu;
          g_reduce<CAP>(a, W, gp);
          for (u32 w = 0; w < n; ++w) a[w] = w == 0 ? 1u : 0u;
          g_reduce<CAP>(a, b, gp);   // (1 mod p)
          g_add<CAP>(W, b, W, gp);
          const u32 groups = (stride + n - 1) / n;
          for (u32 w = 0; w < n; ++w) r[w] = 0;
          for (u32 c = groups; c-- > 0;) {
            for (u32 w = 0; w < n; ++w) a[w] = c * n + w < stride ? q[c * n + w] : 0u;
            g_reduce<CAP>(a, b, gp);
            g_mul<CAP>(r, W, a, gp);
            g_add<CAP>(a, b, r, gp);
          }
        }
        break;
      }
      case OP_ASSERT: {
        has_out = false;
        g_wire_load(T + (size_t)op.a * rec, n, a);
        const bool nz = !g_is_zero(a, n) || g_unreduced_source_is_nonzero<CAP>(op.dst, args, lane_g, lane_valid, gp);
        if (nz && lane_valid) atomicMin(&args.first_fail[lane_g], op.b);
        break;
      }
      default: has_out = false; break;
    }
    if (has_out) g_wire_store(T + (size_t)op.dst * rec, n, r);
  }
}

// KC = 0: any characteristic up to 32 CAP bits, the parameters read from memory; KC > 0: one of KC words (SmallParams)
template <int CAP, int KC>
__global__ __launch_bounds__(256) void replay_generic_kernel(const ReplayArgs args, const GenericParams* __restrict__ gp) {
  if constexpr (KC == 0) {
    replay_generic_body<CAP>(args, gp);
  } else {
    typedef const GenericParams __attribute__((address_space(4))) GpS;   // (wave-uniform: scalar loads)
    GpS* g = (GpS*)(unsigned long long)gp;
    SmallParams<KC> sp;
#pragma unroll
    for (int i = 0; i < KC; ++i) sp.p[i] = g->p[i];
#pragma unroll
    for (int i = 0; i < KC + 2; ++i) sp.mu[i] = g->mu[i];
    sp.pow2_bits = g->pow2_bits;
    replay_generic_body<CAP>(args, &sp);
  }
}

// out[lane][k][n words]: canonical little-endian words of the listed slots (Evaluator::get, evaluator.rs:750-752)
__global__ __launch_bounds__(64) void dump_generic_kernel(const uint4* __restrict__ table, u32 n_slots, const u32* __restrict__ slots,
                                                          u32 n_dump, u32 batch, u32* __restrict__ out, u32 n) {
  const u32 lane = threadIdx.x & 63;
  const u32 lb = blockIdx.y;
  const u32 k = blockIdx.x;
  const u32 lane_g = lb * 64 + lane;
  if (k >= n_dump || lane_g >= batch) return;
  const u32 rec = ((n + 3) / 4) * 64;
  const uint4* T = table + (size_t)lb * n_slots * rec + lane + (size_t)slots[k] * rec;
  u32* o = out + ((size_t)lane_g * n_dump + k) * n;
  for (u32 c = 0; 4 * c < n; ++c) {
    const uint4 v = T[c * 64];
    o[4 * c] = v.x;
    o[4 * c + 1] = v.y;
    if (4 * c + 2 < n) {
      o[4 * c + 2] = v.z;
      o[4 * c + 3] = v.w;
    }
  }
}
#endif  // __HIPCC__

}  // namespace zkgpu
