// GF(2) replay: bit-packed wires, 64 witnesses per 64-bit word.
//
// Device-side equivalent of PlaintextBackend's and/xor/not
// (rust/src/consumers/evaluator.rs:924-938) and of add/mul modulo 2, for
// canonical {0,1} inputs.  The reference keeps one heap BigUint per Boolean
// wire; here one wave moves 4096 witnesses of a wire per instruction.
//
// Wire table: table[lane_block][slot][word 0..63], a lane block = 4096
// witnesses; thread `lane` of a wave owns word `lane`.
// Packed inputs: packed[position][word] for all words of the batch, produced
// from the caller's per-witness byte streams by pack_inputs_kernel.
#pragma once
#include "replay_kernels.hpp"

namespace zkgpu {

struct BoolReplayArgs {
  const TapeOp* ops;
  u32 n_ops;
  u32 ops_per_wave;
  u64* table;
  u32 n_slots;
  u32 batch;
  u32 lb_base;              // first lane block of this launch
  u32 total_words;          // 64 * lane blocks
  const u32* consts;        // 0/1 per constant
  const u64* packed_inst;   // [n_inst][total_words]
  const u64* packed_wit;    // [n_wit][total_words]
  u32* first_fail;
};

// raw[lane][n_vals] bytes -> packed[pos][word]; flags lanes holding a value > 1.
__global__ __launch_bounds__(256) void pack_inputs_kernel(const uint8_t* __restrict__ raw, u32 n_vals, u32 batch,
                                                          u32 total_words, u64* __restrict__ packed,
                                                          u32* __restrict__ lane_flags) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 word = blockIdx.x * (blockDim.x >> 6) + wave;
  if (word >= total_words) return;
  const u32 lane_g = word * 64 + lane;
  const bool valid = lane_g < batch;
  const u32 k0 = blockIdx.y * 256;
  const u32 k1 = min(n_vals, k0 + 256);
  const uint8_t* row = raw + (size_t)lane_g * n_vals;
  bool bad = false;
  for (u32 k = k0; k < k1; ++k) {
    const uint8_t v = valid ? row[k] : 0;
    bad |= v > 1;
    const u64 m = __ballot(v & 1);
    if (lane == 0) packed[(size_t)k * total_words + word] = m;
  }
  if (bad) atomicOr(&lane_flags[lane_g], kLaneFlagNonCanonical);
}

__global__ __launch_bounds__(256) void bool_replay_kernel(const BoolReplayArgs args) {
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane = threadIdx.x & 63;
  const u32 lb = args.lb_base + blockIdx.y;
  const u32 gw = blockIdx.x * (blockDim.x >> 6) + wave;
  const u32 begin = gw * args.ops_per_wave;
  if (begin >= args.n_ops) return;
  const u32 end = min(args.n_ops, begin + args.ops_per_wave);
  const u32 word = lb * 64 + lane;
  u64* __restrict__ T = args.table + (size_t)lb * args.n_slots * 64 + lane;
  // bits of this word that are real witnesses
  const u32 lane0 = word * 64;
  const u64 valid_mask = lane0 >= args.batch ? 0ull
                         : (args.batch - lane0 >= 64 ? ~0ull : ((1ull << (args.batch - lane0)) - 1));
  for (u32 i = begin; i < end; ++i) {
    const TapeOp op = args.ops[i];
    u64 r = 0;
    bool has_out = true;
    switch (op.kind) {
      case OP_XOR: r = T[(size_t)op.a * 64] ^ T[(size_t)op.b * 64]; break;
      case OP_AND: r = T[(size_t)op.a * 64] & T[(size_t)op.b * 64]; break;
      case OP_NOT: r = ~T[(size_t)op.a * 64]; break;
      case OP_COPY: r = T[(size_t)op.a * 64]; break;
      case OP_CONST: r = args.consts[op.a] ? ~0ull : 0ull; break;
      case OP_INSTANCE: r = args.packed_inst[(size_t)op.a * args.total_words + word]; break;
      case OP_WITNESS: r = args.packed_wit[(size_t)op.a * args.total_words + word]; break;
      case OP_ASSERT: {
        has_out = false;
        u64 nz = T[(size_t)op.a * 64] & valid_mask;
        if (__ballot(nz != 0) != 0ull) {  // rare: some witness fails this assert
          while (nz) {
            const u32 bit = __builtin_ctzll(nz);
            nz &= nz - 1;
            atomicMin(&args.first_fail[lane0 + bit], op.b);
          }
        }
        break;
      }
      default: has_out = false; break;
    }
    if (has_out) T[(size_t)op.dst * 64] = r;
  }
}

// out[lane][k] = bit of slot slots[k] (one byte per value)
__global__ __launch_bounds__(64) void bool_dump_slots_kernel(const u64* __restrict__ table, u32 n_slots,
                                                             const u32* __restrict__ slots, u32 n_dump, u32 batch,
                                                             uint8_t* __restrict__ out) {
  const u32 lane_g = blockIdx.y * 64 + threadIdx.x;
  const u32 k = blockIdx.x;
  if (k >= n_dump || lane_g >= batch) return;
  const u32 lb = lane_g / 4096, word = (lane_g % 4096) / 64, bit = lane_g % 64;
  const u64 w = table[((size_t)lb * n_slots + slots[k]) * 64 + word];
  out[(size_t)lane_g * n_dump + k] = (uint8_t)((w >> bit) & 1);
}

}  // namespace zkgpu
