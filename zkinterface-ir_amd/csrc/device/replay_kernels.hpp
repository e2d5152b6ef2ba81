// Tape replay kernels: one lane per witness, wave-uniform tape entries.
//
// Device-side equivalent of the reference's hot loop
// (rust/src/consumers/evaluator.rs:288-301 driving PlaintextBackend
// :892-946): every value-producing ZKBackend call recorded by the host is one
// TapeOp; a wavefront executes an op for 64 witnesses at once.
//
// Wire table layout in HBM (witness-major inside a 64-lane block):
//   table[lane_block][slot][chunk][lane 0..63] of 16-byte chunks
//   chunk c of a value holds 64-bit limbs {2c, 2c+1}; a wave-wide access to
//   one chunk is one contiguous 1 KiB global_load/store_dwordx4.
#pragma once
#include "fp_mont.hpp"

namespace zkgpu {

// wave-uniform program entry on the scalar path (see load_entry_scalar below for the why)
__device__ __forceinline__ TapeOp load_op_scalar(const TapeOp* ops, u32 i) {
  typedef const u32 __attribute__((address_space(4))) cu32;
  cu32* q = (cu32*)(unsigned long long)(ops + __builtin_amdgcn_readfirstlane(i));
  TapeOp op;
  op.dst = q[0];
  op.a = q[1];
  op.b = q[2];
  op.kind = q[3];
  return op;
}

// Workgroup -> (chunk of ops, lane block).  Legacy grid: x = chunk, y = lane block; consecutive
// chunks of one lane block then land on different XCDs (workgroups are dealt round-robin over the 8
// XCDs) and a wire read by two gates of the level is fetched into two L2s.  XCD-aware grid (the launch
// covers a multiple of 8 lane blocks): block b runs on XCD b % 8 and is given lane block
// 8 * group + (b % 8), so each XCD sweeps whole levels of its own lane blocks and the second reader of
// a wire finds it in that XCD's L2.
__device__ __forceinline__ void block_coords(u32 xcd_chunks, u32& chunk, u32& lb_rel) {
  if (xcd_chunks == 0) {
    chunk = blockIdx.x;
    lb_rel = blockIdx.y;
  } else {
    const u32 j = blockIdx.x >> 3;
    const u32 g = j / xcd_chunks;
    chunk = j - g * xcd_chunks;
    lb_rel = g * 8 + (blockIdx.x & 7);
  }
}

template <int N>
struct Layout {
  static constexpr int kChunks = (N + 3) / 4;          // 16-byte chunks (last may be half used)
  static constexpr int kRecord = kChunks * 64;         // uint4 per (lane block, slot)
};

template <int N>
__device__ __forceinline__ Fp<N> wire_load(const uint4* __restrict__ rec) {
  Fp<N> r;
#pragma unroll
  for (int c = 0; c < Layout<N>::kChunks; ++c) {
    if constexpr (N % 4 == 2) {
      if (c == Layout<N>::kChunks - 1) {
        const uint2 v = *reinterpret_cast<const uint2*>(&rec[c * 64]);
        r.w[4 * c] = v.x;
        r.w[4 * c + 1] = v.y;
        continue;
      }
    }
    const uint4 v = rec[c * 64];
    r.w[4 * c] = v.x;
    r.w[4 * c + 1] = v.y;
    r.w[4 * c + 2] = v.z;
    r.w[4 * c + 3] = v.w;
  }
  return r;
}

template <int N>
__device__ __forceinline__ void wire_store(uint4* __restrict__ rec, const Fp<N>& r) {
#pragma unroll
  for (int c = 0; c < Layout<N>::kChunks; ++c) {
    if constexpr (N % 4 == 2) {
      if (c == Layout<N>::kChunks - 1) {
        *reinterpret_cast<uint2*>(&rec[c * 64]) = make_uint2(r.w[4 * c], r.w[4 * c + 1]);
        continue;
      }
    }
    rec[c * 64] = make_uint4(r.w[4 * c], r.w[4 * c + 1], r.w[4 * c + 2], r.w[4 * c + 3]);
  }
}

// Strands (replay_strand_kernel): a value that never leaves its strand lives in the workgroup's LDS, value k at
// [k][chunk][lane] in 16-byte chunks like a record of the wire table (a wave's access to a chunk is 64 consecutive 16-byte
// words: conflict-free ds_read_b128 / ds_write_b128).  A slot number with kSlotInLds names such a value; slot numbers are
// wave-uniform (they come out of the program entry, on the scalar path), so the choice is a scalar branch.
extern __shared__ __attribute__((aligned(16))) uint4 zk_strand_lds[];

template <int N>
__device__ __forceinline__ Fp<N> lds_value_load(u32 k, u32 lane) {
  Fp<N> r;
  const u32 base = k * Layout<N>::kRecord + lane;
#pragma unroll
  for (int c = 0; c < Layout<N>::kChunks; ++c) {
    const uint4 v = zk_strand_lds[base + c * 64];
    r.w[4 * c] = v.x;
    r.w[4 * c + 1] = v.y;
    if (4 * c + 2 < N) {
      r.w[4 * c + 2] = v.z;
      r.w[4 * c + 3] = v.w;
    }
  }
  return r;
}
template <int N>
__device__ __forceinline__ void lds_value_store(u32 k, u32 lane, const Fp<N>& r) {
  const u32 base = k * Layout<N>::kRecord + lane;
#pragma unroll
  for (int c = 0; c < Layout<N>::kChunks; ++c)
    zk_strand_lds[base + c * 64] = make_uint4(r.w[4 * c], r.w[4 * c + 1], 4 * c + 2 < N ? r.w[4 * c + 2] : 0u, 4 * c + 2 < N ? r.w[4 * c + 3] : 0u);
}
// the value of a slot: out of the wire table, or (LDS: a strand's kernel) out of the workgroup's LDS
template <int N, bool LDS>
__device__ __forceinline__ Fp<N> slot_load(const uint4* __restrict__ T, u32 slot, u32 lane) {
  if constexpr (LDS) {
    if (slot & kSlotInLds) return lds_value_load<N>(slot & ~kSlotInLds, lane);
  }
  return wire_load<N>(T + (size_t)slot * Layout<N>::kRecord);
}
template <int N, bool LDS>
__device__ __forceinline__ void slot_store(uint4* __restrict__ T, u32 slot, u32 lane, const Fp<N>& r) {
  if constexpr (LDS) {
    if (slot & kSlotInLds) {
      lds_value_store<N>(slot & ~kSlotInLds, lane, r);
      return;
    }
  }
  wire_store<N>(T + (size_t)slot * Layout<N>::kRecord, r);
}

// One input value of a lane: the low N words of its `stride_words`-word slot (the slot is narrower than N only for a
// value carried over from a narrower field; it is wider in a session of several fields, whose input buffers have the
// width of the widest one).  too_wide: the value has bits above the N words -- it is >= R > p and its residue cannot be
// taken by to_mont: the caller flags the lane.
template <int N>
__device__ __forceinline__ Fp<N> input_load(const void* __restrict__ base, u32 lane_g, u32 n_vals, u32 idx, bool valid,
                                            u32 stride_words, bool& too_wide) {
  Fp<N> r;
  too_wide = false;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = 0;
  if (valid) {
    const u32* p = reinterpret_cast<const u32*>(base) + ((size_t)lane_g * n_vals + idx) * stride_words;
    if (stride_words == (u32)N) {   // the usual case (wave-uniform): N unconditional loads, which hipcc merges into dwordx4
#pragma unroll
      for (int i = 0; i < N; ++i) r.w[i] = p[i];
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i)
        if ((u32)i < stride_words) r.w[i] = p[i];
      u32 hi = 0;
      for (u32 i = N; i < stride_words; ++i) hi |= p[i];
      too_wide = hi != 0;
    }
  }
  return r;
}

// the raw value of input stream 0 (instance), 1 (witness) or 2 (carry) at `position`, or of raw constant `position` (3)
typedef const InputAux __attribute__((address_space(4))) InputAuxS;   // (scalar loads)
template <int N, class Args>
__device__ __forceinline__ Fp<N> stream_load(u32 stream, u32 position, const Args& args, u32 lane_g, bool lane_valid, bool& too_wide) {
  InputAuxS* aux = (InputAuxS*)(unsigned long long)args.aux;
  if (stream == 3) {   // a constant kept as the integer it is (the same for every lane)
    Fp<N> r;
    const u32* c = args.consts + (size_t)(aux->raw_const_base + position) * N;
#pragma unroll
    for (int i = 0; i < N; ++i) r.w[i] = c[i];
    too_wide = false;
    return r;
  }
  if (stream == 2) return input_load<N>(aux->carry, lane_g, aux->n_carry, position, lane_valid, aux->carry_words, too_wide);
  return input_load<N>(stream ? args.wit : args.inst, lane_g, stream ? args.n_wit : args.n_inst, position, lane_valid,
                       aux->in_stride_words, too_wide);
}

// Input positions whose value must be canonical (mode 0xFF in Schedule::strict_instance / strict_witness: it reaches an
// integer bit operation or Evaluator::get unreduced): a value >= p there flags the lane.
__device__ __forceinline__ bool position_is_strict(const uint8_t* __restrict__ modes, u32 position) {
  return modes[position] == 0xFF;
}

// The operand of an assert_zero / not that a constant, instance or witness value reaches through copies alone is, for the
// reference, the UNREDUCED integer (evaluator.rs:862-864,896-906,932-946): a value >= p is not zero whatever its residue.
// `code` (schedule.cpp track_unreduced_values): 0 = no such source, 1 = a constant >= p, 2 + 4 * position + stream (0
// instance, 1 witness, 2 carried from the previous field segment) = an input, whose raw value is tested here beside the wire.
template <int N, class Args>
__device__ __forceinline__ bool unreduced_source_is_nonzero(u32 code, const Args& args, u32 lane_g, bool lane_valid,
                                                            const FieldParams& fp) {
  if (code < 2) return code == 1;
  const u32 q = code - 2;
  bool too_wide;
  const Fp<N> raw = stream_load<N>(q & 3, q >> 2, args, lane_g, lane_valid, too_wide);
  return lane_valid && (too_wide || fp_geq_p<N>(raw, fp));
}

// Words [c N, (c + 1) N) of an input value of `stride_words` words, zero-extended: a value wider than the field's limbs
// (a session whose fields differ in width hands its inputs over in the width of the widest, and a wire carried over from
// a wider field is as wide as that field) is reduced limb group by limb group (input_op).
template <int N>
__device__ __forceinline__ Fp<N> input_chunk(const void* __restrict__ base, u32 lane_g, u32 n_vals, u32 idx, bool valid, u32 stride_words, u32 c) {
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = 0;
  if (valid) {
    const u32* p = reinterpret_cast<const u32*>(base) + ((size_t)lane_g * n_vals + idx) * stride_words;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (c * N + i < stride_words) r.w[i] = p[c * N + i];
  }
  return r;
}

// An input op of either replay kernel: load, flag the lane where the residue will not do, and return the Montgomery form
// of the value's residue.  The reference keeps inputs unreduced (evaluator.rs:862-864,940-946) and arithmetic reduces
// them, `(a + b) % m`, however large they are: at a position that only arithmetic reads (mode 0x00, or 0x01 / 0x02 whose
// zero tests look at the raw input themselves) a value wider than the N words is REDUCED -- Horner over its groups of N
// words, R = 2^(32 N): x R mod p = ((..(top R + next) R + ..) R + low) R, i.e. to_mont(group) added to the running value
// times R (a Montgomery product by R^2) -- and only where its bits matter (0xFF: it reaches an integer bit operation or
// Evaluator::get as it is; 0x03: and / xor read the raw input, which their N-word operand cannot hold) is the lane flagged.
// (input_op in its parts: the lane is flagged where the raw words of position `position` of stream `stream` must not be
// what they are -- the mode byte comes from memory through two dependent loads --, and the conversion)
template <int N, class Args>
__device__ __forceinline__ void input_check(u32 stream, u32 position, const Fp<N>& raw, bool too_wide, const Args& args, u32 lane_g,
                                            bool lane_valid, const FieldParams& fp) {
  InputAuxS* aux = (InputAuxS*)(unsigned long long)args.aux;
  const uint8_t* modes = stream == 0 ? aux->strict_inst : stream == 1 ? aux->strict_wit : aux->strict_carry;
  const u32 mode = modes[position];
  if (lane_valid && ((too_wide && (mode == 0xFF || mode == 0x03)) || (mode == 0xFF && fp_geq_p<N>(raw, fp))))
    atomicOr(&args.lane_flags[lane_g], kLaneFlagNonCanonical);
}
template <int N, class Args>
__device__ __forceinline__ Fp<N> input_convert(u32 stream, u32 position, const Fp<N>& raw, bool too_wide, const Args& args, u32 lane_g,
                                               bool lane_valid, const FieldParams& fp) {
  InputAuxS* aux = (InputAuxS*)(unsigned long long)args.aux;
  if (__ballot(too_wide && lane_valid) == 0ull) return fp_to_mont<N>(raw, fp);   // of any value < R: the Montgomery form of its residue
  // some lane holds a value of more than N words (rare: one pass over the groups for the whole wave)
  const void* base = stream == 2 ? (const void*)aux->carry : (const void*)(stream ? args.wit : args.inst);
  const u32 n_vals = stream == 2 ? aux->n_carry : (stream ? args.n_wit : args.n_inst);
  const u32 stride = stream == 2 ? aux->carry_words : aux->in_stride_words;
  const u32 groups = (stride + N - 1) / N;
  Fp<N> r2;
#pragma unroll
  for (int i = 0; i < N; ++i) r2.w[i] = fp.r2[i];
  Fp<N> acc = fp_to_mont<N>(input_chunk<N>(base, lane_g, n_vals, position, lane_valid, stride, groups - 1), fp);
  for (u32 c = groups - 1; c-- > 0;)
    acc = fp_add<N>(fp_mul<N>(acc, r2, fp), fp_to_mont<N>(input_chunk<N>(base, lane_g, n_vals, position, lane_valid, stride, c), fp), fp);
  return acc;
}

template <int N, class Args>
__device__ __forceinline__ Fp<N> input_op(u32 kind, u32 position, const Args& args, u32 lane_g, bool lane_valid, const FieldParams& fp) {
  const u32 stream = kind == OP_INSTANCE ? 0u : kind == OP_WITNESS ? 1u : 2u;
  bool too_wide;
  const Fp<N> raw = stream_load<N>(stream, position, args, lane_g, lane_valid, too_wide);
  input_check<N>(stream, position, raw, too_wide, args, lane_g, lane_valid, fp);
  return input_convert<N>(stream, position, raw, too_wide, args, lane_g, lane_valid, fp);
}

// An operand of an integer bit operation (`and` / `xor` over an odd field, evaluator.rs:924-933): the canonical integer of
// a wire -- or, where the wire is an input the relation has only copied, the RAW value of that input (reference
// kOperandIsSource | code, the code of unreduced_source_is_nonzero): PlaintextBackend keeps inputs unreduced and the bit
// operations work on those bits.  A raw value that does not fit the limbs flags the lane.
template <int N, class Args, bool LDS = false>
__device__ __forceinline__ Fp<N> bit_operand(u32 ref, const uint4* __restrict__ T, const Args& args, u32 lane_g, bool lane_valid,
                                             const FieldParams& fp) {
  if (ref & kOperandIsSource) {
    const u32 q = (ref & ~kOperandIsSource) - 2;
    bool too_wide;
    const Fp<N> raw = stream_load<N>(q & 3, q >> 2, args, lane_g, lane_valid, too_wide);
    if (lane_valid && too_wide) atomicOr(&args.lane_flags[lane_g], kLaneFlagNonCanonical);
    return raw;
  }
  return fp_from_mont<N>(slot_load<N, LDS>(T, ref, lane_g & 63), fp);
}
template <int N, class Args, bool LDS = false>
__device__ __forceinline__ Fp<N> bit_operation(u32 kind, u32 ref_a, u32 ref_b, const uint4* __restrict__ T, const Args& args, u32 lane_g,
                                               bool lane_valid, const FieldParams& fp) {
  const Fp<N> x = bit_operand<N, Args, LDS>(ref_a, T, args, lane_g, lane_valid, fp), y = bit_operand<N, Args, LDS>(ref_b, T, args, lane_g, lane_valid, fp);
  Fp<N> r;
#pragma unroll
  for (int i = 0; i < N; ++i) r.w[i] = kind == OP_AND ? (x.w[i] & y.w[i]) : (x.w[i] ^ y.w[i]);
  return fp_to_mont<N>(r, fp);   // of any value below R: the Montgomery form of its residue
}

// One wave = 64 witnesses x `ops_per_wave` consecutive tape ops.
// PIPE: operands of op i+1 are requested before op i is computed; legal only
// when the ops of one wave are mutually independent (a level of the schedule).
// BITOPS: the and / xor arms of PlaintextBackend over an odd field (two from_mont + one to_mont each) are only
// compiled into the instantiation the host picks for programs that contain them -- they cost the common
// instantiation 19 VGPRs (83 -> 102) it never uses.
template <int N, bool PIPE, bool BITOPS = true>
__global__ __launch_bounds__(256) void replay_kernel(const ReplayArgs args, const FieldParams fp) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  u32 chunk, lb_rel;
  block_coords(args.xcd_chunks, chunk, lb_rel);
  const u32 lb = args.lb_base + lb_rel;
  const u32 gw = chunk * (blockDim.x >> 6) + wave;
  const u32 begin = gw * args.ops_per_wave;
  if (begin >= args.n_ops) return;
  const u32 end = min(args.n_ops, begin + args.ops_per_wave);
  const u32 lane_g = lb * 64 + lane;
  const bool lane_valid = lane_g < args.batch;
  uint4* __restrict__ T = args.table + (size_t)lb * args.n_slots * Layout<N>::kRecord + lane;
  constexpr int REC = Layout<N>::kRecord;

  // (and / xor fetch their own operands: they may name an input instead of a wire, bit_operand)
  auto needs_a = [](u32 k) { return k != OP_CONST && k != OP_INSTANCE && k != OP_WITNESS && k != OP_CARRY && k != OP_NOP && k != OP_AND && k != OP_XOR; };  // OP_NZ reads a
  auto needs_b = [](u32 k) { return k == OP_ADD || k == OP_MUL; };

  TapeOp op = load_op_scalar(args.ops, begin);
  Fp<N> a, b;
  if (needs_a(op.kind)) a = wire_load<N>(T + (size_t)op.a * REC);
  if (needs_b(op.kind)) b = wire_load<N>(T + (size_t)op.b * REC);

  for (u32 i = begin; i < end; ++i) {
    TapeOp nop;
    Fp<N> na, nb;
    nop.kind = OP_NOP;
    if (PIPE && i + 1 < end) {
      nop = load_op_scalar(args.ops, i + 1);
      if (needs_a(nop.kind)) na = wire_load<N>(T + (size_t)nop.a * REC);
      if (needs_b(nop.kind)) nb = wire_load<N>(T + (size_t)nop.b * REC);
    }
    Fp<N> r;
    bool has_out = true;
    switch (op.kind) {
      case OP_ADD: r = fp_add<N>(a, b, fp); break;
      case OP_MUL: r = fp_mul<N>(a, b, fp); break;
      case OP_ADDC: r = fp_add<N>(a, fp_load_const<N>(args.consts + (size_t)op.b * N), fp); break;
      case OP_MULC: r = fp_mul<N>(a, fp_load_const<N>(args.consts + (size_t)op.b * N), fp); break;
      case OP_COPY: r = a; break;
      case OP_NZ: r = fp_nonzero_indicator<N>(a, fp); break;
      case OP_AND:   // integer bit ops of PlaintextBackend over an odd field
      case OP_XOR:
        if constexpr (BITOPS) r = bit_operation<N>(op.kind, op.a, op.b, T, args, lane_g, lane_valid, fp); else has_out = false;
        break;
      case OP_NOT:   // op.b: the unreduced source behind the operand, if any
        r = fp_indicator<N>(fp_is_zero<N>(a) && !unreduced_source_is_nonzero<N>(op.b, args, lane_g, lane_valid, fp), fp);
        break;
      case OP_CONST: r = fp_load_const<N>(args.consts + (size_t)op.a * N); break;
      case OP_INSTANCE:
      case OP_WITNESS:
      case OP_CARRY: r = input_op<N>(op.kind, op.a, args, lane_g, lane_valid, fp); break;
      case OP_ASSERT: {
        has_out = false;
        const bool nz = !fp_is_zero<N>(a) || unreduced_source_is_nonzero<N>(op.dst, args, lane_g, lane_valid, fp);
        if (__ballot(nz && lane_valid) != 0ull) {
          if (nz && lane_valid) atomicMin(&args.first_fail[lane_g], op.b);
        }
        break;
      }
      default: has_out = false; break;
    }
    if (has_out) wire_store<N>(T + (size_t)op.dst * REC, r);
    if (PIPE) {
      op = nop;
      a = na;
      b = nb;
    } else if (i + 1 < end) {
      // sequential segment (e.g. the square-and-multiply ladder of a Switch weight,
      // evaluator.rs:801-820): the next op usually consumes the value just produced -- forward it
      // from registers instead of reading the wire table back.
      const u32 produced = has_out ? op.dst : 0xFFFFFFFFu;
      op = load_op_scalar(args.ops, i + 1);
      if (needs_a(op.kind)) a = (op.a == produced) ? r : wire_load<N>(T + (size_t)op.a * REC);
      if (needs_b(op.kind)) b = (op.b == produced) ? r : wire_load<N>(T + (size_t)op.b * REC);
    }
  }
}

// A program entry is the same for the whole wave: fetch it on the scalar path (s_load through the scalar cache
// into SGPRs).  Left to itself hipcc reads it with a vector load + six v_readfirstlane, which waits in vmcnt in
// front of the operand gathers and keeps the entry in VGPRs (80 -> 74 registers, 9.33 -> 9.25 ms on C2).
// The program is never written while a replay runs.
__device__ __forceinline__ TapeOp2 load_entry_scalar(const TapeOp2* ops, u32 i) {
  typedef const u32 __attribute__((address_space(4))) cu32;
  cu32* q = (cu32*)(unsigned long long)(ops + __builtin_amdgcn_readfirstlane(i));
  TapeOp2 op;
  op.dst = q[0];
  op.kind = q[1];
  op.a0 = q[2];
  op.a1 = q[3];
  op.b0 = q[4];
  op.b1 = q[5];
  op.pad0 = q[6];
  op.pad1 = q[7];
  return op;
}

// ... without the two words only a pair entry has (fetched ahead of their use by the strand kernel, which keeps an entry
// in SGPRs while another one runs), and those two words
__device__ __forceinline__ TapeOp2 load_entry_scalar6(const TapeOp2* ops, u32 i) {
  typedef const u32 __attribute__((address_space(4))) cu32;
  cu32* q = (cu32*)(unsigned long long)(ops + __builtin_amdgcn_readfirstlane(i));
  TapeOp2 op;
  op.dst = q[0];
  op.kind = q[1];
  op.a0 = q[2];
  op.a1 = q[3];
  op.b0 = q[4];
  op.b1 = q[5];
  op.pad0 = op.pad1 = 0;
  return op;
}
__device__ __forceinline__ void load_entry_pair_words(const TapeOp2* ops, u32 i, TapeOp2& op) {
  typedef const u32 __attribute__((address_space(4))) cu32;
  cu32* q = (cu32*)(unsigned long long)(ops + __builtin_amdgcn_readfirstlane(i));
  op.pad0 = q[6];
  op.pad1 = q[7];
}

// Replay of the fused schedule: an Add/Mul operand may be `add(a0,a1)` / `mul(a0,a1)` evaluated in
// registers -- the absorbed producer's value never goes to the wire table (one 32-B store and one
// 32-B load less per fused pair).  All gathers of an op are issued before the arithmetic.
//
// Three instantiations per field width, chosen per launch by the host (the scheduler puts the Add/Mul entries
// of a level first):
//   kFusedHot  -- Add/Mul entries only (with fused producers and pair entries): what a wide level of an
//                 arithmetic relation consists of.  No other arm is compiled in, so the register allocation is
//                 that of the Add/Mul body alone (<= 80 VGPRs at 8 words: 6 waves per SIMD).
//   kFusedMisc -- every kind except the integer bit operations over an odd field (inputs, constants, copies,
//                 AddConstant/MulConstant, AssertZero, the `x != 0` indicator, and Add/Mul for sequential segments).
//   kFusedAll  -- kFusedMisc + and / xor over an odd field (two from_mont + one to_mont each, evaluator.rs:924-933).
template <int N, bool LDS = false>
__device__ __forceinline__ void fused_addmul(const TapeOp2& op, uint4* __restrict__ T, const FieldParams& fp, u32 lane = 0) {
  const u32 kind = op.kind & 0xFF, ea = (op.kind >> 8) & 3, eb = (op.kind >> 10) & 3;
  // (in the order of their use: where an operand may come from LDS or from the wire table -- a strand -- the wait in front
  // of its first use covers every load issued before it on either path, so the operands of the first inner operation go
  // first and what only the last operation reads goes last: its fetch from HBM then overlaps the inner product)
  Fp<N> x0 = slot_load<N, LDS>(T, op.a0, lane), x1, y0, y1;
  if (ea) x1 = slot_load<N, LDS>(T, op.a1, lane);
  y0 = slot_load<N, LDS>(T, op.b0, lane);
  if (eb) y1 = slot_load<N, LDS>(T, op.b1, lane);
  u32 pv[N];   // the words of p in VGPRs, once per entry: every carry chain below subtracts them (fp_mont.hpp)
  if constexpr (N <= 12) p_words_resident<N>(pv, fp);
  else p_words<N>(pv, fp);   // (beyond 384 bits the kernel is short of registers: let hipcc rematerialise them)
  if (ea) x0 = ea == 1 ? fp_add<N>(x0, x1, fp, pv) : fp_mul<N>(x0, x1, fp, pv);
  if (eb) y0 = eb == 1 ? fp_add<N>(y0, y1, fp, pv) : fp_mul<N>(y0, y1, fp, pv);
  Fp<N> r = kind == OP_ADD ? fp_add<N>(x0, y0, fp, pv) : fp_mul<N>(x0, y0, fp, pv);
  u32 dst_slot = op.dst;
  const u32 pair = (op.kind >> 12) & 3;
  if (pair) {
    // a second gate of the same level fed by the shared producer X: store the first result, fetch the second
    // gate's other operand into registers the first no longer needs, and let the common store write it
    slot_store<N, LDS>(T, dst_slot, lane, r);
    y0 = slot_load<N, LDS>(T, op.pad1, lane);
    r = pair == 1 ? fp_add<N>(x0, y0, fp, pv) : fp_mul<N>(x0, y0, fp, pv);
    dst_slot = op.pad0;
  }
  slot_store<N, LDS>(T, dst_slot, lane, r);
}

// The chain of a strand: an Add/Mul entry (not a pair) whose operands and result all live in LDS.  fused_addmul reaches
// every operand through a branch of its own (LDS or wire table), and hipcc closes each with a wait: the four loads of an
// entry then run one after the other, ~100 cycles each for a wave that has its SIMD to itself.  Here all four are issued
// together (an operand the entry does not have reads its neighbour again) and waited for once.
template <int N>
__device__ __forceinline__ bool strand_entry_in_lds(const TapeOp2& op) {
  const u32 kind = op.kind & 0xFF, ea = (op.kind >> 8) & 3, eb = (op.kind >> 10) & 3, pair = (op.kind >> 12) & 3;
  const u32 all = op.dst & op.a0 & op.b0 & (ea ? op.a1 : ~0u) & (eb ? op.b1 : ~0u);
  return (kind == OP_ADD || kind == OP_MUL) && pair == 0 && (all & kSlotInLds) != 0;
}
template <int N>
__device__ __forceinline__ void strand_addmul_lds(const TapeOp2& op, const FieldParams& fp, u32 lane) {
  const u32 kind = op.kind & 0xFF, ea = (op.kind >> 8) & 3, eb = (op.kind >> 10) & 3;
  Fp<N> x0 = lds_value_load<N>(op.a0 & ~kSlotInLds, lane);
  const Fp<N> x1 = lds_value_load<N>((ea ? op.a1 : op.a0) & ~kSlotInLds, lane);
  Fp<N> y0 = lds_value_load<N>(op.b0 & ~kSlotInLds, lane);
  const Fp<N> y1 = lds_value_load<N>((eb ? op.b1 : op.b0) & ~kSlotInLds, lane);
  u32 pv[N];
  if constexpr (N <= 12) p_words_resident<N>(pv, fp);
  else p_words<N>(pv, fp);
  if (ea) x0 = ea == 1 ? fp_add<N>(x0, x1, fp, pv) : fp_mul<N>(x0, x1, fp, pv);
  if (eb) y0 = eb == 1 ? fp_add<N>(y0, y1, fp, pv) : fp_mul<N>(y0, y1, fp, pv);
  const Fp<N> r = kind == OP_ADD ? fp_add<N>(x0, y0, fp, pv) : fp_mul<N>(x0, y0, fp, pv);
  lds_value_store<N>(op.dst & ~kSlotInLds, lane, r);
}

// one entry of any kind (the body of the kFusedMisc / kFusedAll instantiations)
template <int N, int CLS, bool LDS = false>
__device__ __forceinline__ void fused_entry(const TapeOp2& op, uint4* __restrict__ T, const ReplayArgs2& args, u32 lane_g,
                                            bool lane_valid, const FieldParams& fp) {
  const u32 kind = op.kind & 0xFF, lane = lane_g & 63;
  Fp<N> r;
  bool has_out = true;
  switch (kind) {
    case OP_ADD:
    case OP_MUL: fused_addmul<N, LDS>(op, T, fp, lane); has_out = false; break;
    case OP_ADDC: r = fp_add<N>(slot_load<N, LDS>(T, op.a0, lane), fp_load_const<N>(args.consts + (size_t)op.b0 * N), fp); break;
    case OP_MULC: r = fp_mul<N>(slot_load<N, LDS>(T, op.a0, lane), fp_load_const<N>(args.consts + (size_t)op.b0 * N), fp); break;
    case OP_COPY: r = slot_load<N, LDS>(T, op.a0, lane); break;
    case OP_NZ: r = fp_nonzero_indicator<N>(slot_load<N, LDS>(T, op.a0, lane), fp); break;
    case OP_AND:
    case OP_XOR:
      if constexpr (CLS == kFusedAll) r = bit_operation<N, ReplayArgs2, LDS>(kind, op.a0, op.b0, T, args, lane_g, lane_valid, fp);
      else has_out = false;
      break;
    case OP_NOT:   // op.a1: the unreduced source behind the operand, if any
      r = fp_indicator<N>(fp_is_zero<N>(slot_load<N, LDS>(T, op.a0, lane)) &&
                              !unreduced_source_is_nonzero<N>(op.a1, args, lane_g, lane_valid, fp), fp);
      break;
    case OP_CONST: r = fp_load_const<N>(args.consts + (size_t)op.a0 * N); break;
    case OP_INSTANCE:
    case OP_WITNESS:
    case OP_CARRY: r = input_op<N>(kind, op.a0, args, lane_g, lane_valid, fp); break;
    case OP_INPUT_RAW:    // (strands: the fetch from HBM in one level ...
      if constexpr (LDS) {
        // a buffer whose values are wider than this field's N words (a session of several fields) is fetched by the
        // conversion itself, the general way
        if (args.op_stride == 1u /* (a strand launch: the buffers hold N-word values, device/args.hpp) */) {
          bool too_wide;
          r = stream_load<N>(op.a1, op.a0, args, lane_g, lane_valid, too_wide);
          input_check<N>(op.a1, op.a0, r, too_wide, args, lane_g, lane_valid, fp);   // (here, off the chain: the flag only accumulates)
        } else {
          has_out = false;
        }
      } else {
        has_out = false;
      }
      break;
    case OP_INPUT_CONV:   // ... the conversion in a later one)
      if constexpr (LDS) {
        bool too_wide = false;
        Fp<N> raw;
        if (args.op_stride == 1u /* (a strand launch: the buffers hold N-word values, device/args.hpp) */) {
          raw = slot_load<N, LDS>(T, op.a0, lane);   // checked by the entry that fetched it
        } else {
          raw = stream_load<N>(op.a1, op.b0, args, lane_g, lane_valid, too_wide);
          input_check<N>(op.a1, op.b0, raw, too_wide, args, lane_g, lane_valid, fp);
        }
        r = input_convert<N>(op.a1, op.b0, raw, too_wide, args, lane_g, lane_valid, fp);
      } else {
        has_out = false;
      }
      break;
    case OP_ASSERT: {
      has_out = false;
      const bool nz = !fp_is_zero<N>(slot_load<N, LDS>(T, op.a0, lane)) ||
                      unreduced_source_is_nonzero<N>(op.a1, args, lane_g, lane_valid, fp);
      if (__ballot(nz && lane_valid) != 0ull) {
        if (nz && lane_valid) atomicMin(&args.first_fail[lane_g], op.b0);
      }
      break;
    }
    default: has_out = false; break;
  }
  if (has_out) slot_store<N, LDS>(T, op.dst, lane, r);
}

template <int N, int CLS>
__global__ __launch_bounds__(256) void replay_fused_kernel(const ReplayArgs2 args, const FieldParams fp) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  u32 chunk, lb_rel;
  block_coords(args.xcd_chunks, chunk, lb_rel);
  const u32 lb = args.lb_base + lb_rel;
  // Ops of a workgroup: consecutive per wave (op_stride 1), or interleaved over its 4 waves (op_stride 4): the
  // neighbours of the shared-operand order then run at the same time in neighbouring waves.
  const u32 stride = args.op_stride;
  const u32 begin = stride == 1 ? (chunk * 4 + wave) * args.ops_per_wave : chunk * 4 * args.ops_per_wave + wave;
  if (begin >= args.n_ops) return;
  const u32 end = min(args.n_ops, begin + args.ops_per_wave * stride);
  uint4* __restrict__ T = args.table + (size_t)lb * args.n_slots * Layout<N>::kRecord + lane;
  if constexpr (CLS == kFusedHot) {
    for (u32 i = begin; i < end; i += stride) fused_addmul<N>(load_entry_scalar(args.ops, i), T, fp);
  } else {
    const u32 lane_g = lb * 64 + lane;
    const bool lane_valid = lane_g < args.batch;
    for (u32 i = begin; i < end; i += stride) fused_entry<N, CLS>(load_entry_scalar(args.ops, i), T, args, lane_g, lane_valid, fp);
  }
}

// Strands: a run of consecutive NARROW levels (fewer entries than pay for a launch of their own -- the dependency
// chains of a structured relation: one iteration of a loop feeding the next) is walked by ONE workgroup per lane block,
// level by level, its four waves sharing the entries of a level, with a workgroup barrier between levels.  The barrier
// also orders the memory accesses: the waves of a workgroup share their CU's L1, so workgroup scope is all it takes for
// wave B to see what wave A stored to the wire table before the barrier.  One launch instead of one per level: the
// reference's example relation chained 1,408 times is 7,046 levels.
template <int N, int CLS>
__global__ __launch_bounds__(256) void replay_strand_kernel(const ReplayArgs2 args, const u32* __restrict__ level_ptr, u32 n_levels,
                                                            const FieldParams fp) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 lb = args.lb_base + blockIdx.x;
  const u32 lane_g = lb * 64 + lane;
  const bool lane_valid = lane_g < args.batch;
  uint4* __restrict__ T = args.table + (size_t)lb * args.n_slots * Layout<N>::kRecord + lane;
  typedef const u32 __attribute__((address_space(4))) cu32;
  cu32* lp = (cu32*)(unsigned long long)level_ptr;
  u32 b = lp[0], e = lp[1];
#ifdef ZKGPU_STRAND_STAMPS   // developer build (tools/strand_stamps.py): where the time of a level goes
  unsigned long long* const stamps = ((InputAuxS*)(unsigned long long)args.aux)->stamps;
  const bool stamp = stamps != nullptr && blockIdx.x == 0 && n_levels >= kStampLevels;
#endif
  // The entry a wave runs next -- its next one of this level, or its first of the next level -- is fetched (scalar loads)
  // while the current one runs or the barrier is waited for: a level of a dependency chain is one entry long, and the
  // ~150-250 cycles of an entry fetch were paid once per level in front of it.
  TapeOp2 next_op = {};
  if (b + wave < e) next_op = load_entry_scalar6(args.ops, b + wave);
  for (u32 l = 0; l < n_levels; ++l) {
    const u32 e2 = l + 1 < n_levels ? lp[l + 2] : e;   // the end of the next level (none: nothing to fetch there)
#ifdef ZKGPU_STRAND_STAMPS
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (stamp) t0 = t1 = t2 = __builtin_readcyclecounter();
#endif
    if (b + wave >= e && e + wave < e2) next_op = load_entry_scalar6(args.ops, e + wave);   // idle in this level
    for (u32 i = b + wave; i < e; i += 4) {
      TapeOp2 op = next_op;
      if ((op.kind >> 12) & 3) load_entry_pair_words(args.ops, i, op);   // (a pair entry: rare in a strand)
      const u32 j = i + 4 < e ? i + 4 : e + wave;
      if (i + 4 < e || j < e2) next_op = load_entry_scalar6(args.ops, j);
#ifdef ZKGPU_STRAND_STAMPS
      if (stamp) t1 = __builtin_readcyclecounter();
#endif
      if (strand_entry_in_lds<N>(op)) strand_addmul_lds<N>(op, fp, lane);
      else fused_entry<N, CLS, true>(op, T, args, lane_g, lane_valid, fp);
    }
#ifdef ZKGPU_STRAND_STAMPS
    if (stamp) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      t2 = __builtin_readcyclecounter();
    }
#endif
    b = e;
    e = e2;
    __syncthreads();   // every wave of the workgroup reaches it once per level (the level bounds are wave-uniform)
#ifdef ZKGPU_STRAND_STAMPS
    if (stamp && l < kStampLevels && lane == 0) {
      unsigned long long* q = stamps + ((size_t)l * 4 + wave) * 4;
      q[0] = t0;
      q[1] = t1;
      q[2] = t2;
      q[3] = __builtin_readcyclecounter();
    }
#endif
  }
}

// Dump one slot of every lane back to canonical little-endian words:
// out[lane][N] (parity tests; evaluator.rs:750-752 `Evaluator::get`).
template <int N>
__global__ __launch_bounds__(64) void dump_slots_kernel(const uint4* __restrict__ table, u32 n_slots,
                                                        const u32* __restrict__ slots, u32 n_dump, u32 batch,
                                                        u32* __restrict__ out, const FieldParams fp) {
  const u32 lane = threadIdx.x & 63;
  const u32 lb = blockIdx.y;
  const u32 k = blockIdx.x;
  const u32 lane_g = lb * 64 + lane;
  if (k >= n_dump || lane_g >= batch) return;
  const uint4* T = table + (size_t)lb * n_slots * Layout<N>::kRecord + lane;
  Fp<N> v = wire_load<N>(T + (size_t)slots[k] * Layout<N>::kRecord);
  v = fp_from_mont<N>(v, fp);
  u32* o = out + ((size_t)lane_g * n_dump + k) * N;
#pragma unroll
  for (int i = 0; i < N; ++i) o[i] = v.w[i];
}

}  // namespace zkgpu
