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

// Final verdict reduction (every field; lives here because this header is compiled exactly once): satisfied = lanes with no failing assert and no flag.
// counts[0] += satisfied, counts[1] += failed (u64 each), one atomic per wave.
__global__ __launch_bounds__(256) void verdict_kernel(const u32* __restrict__ first_fail,
                                                       const u32* __restrict__ lane_flags, u32 batch,
                                                       unsigned long long* __restrict__ counts) {
  const u32 lane_g = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = lane_g < batch;
  const bool ok = valid && first_fail[lane_g] == kNoFail && lane_flags[lane_g] == 0;
  const unsigned long long okm = __ballot(ok);
  const unsigned long long vm = __ballot(valid);
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long n_ok = __popcll(okm);
    const unsigned long long n_v = __popcll(vm);
    if (n_v) {
      atomicAdd(&counts[0], n_ok);
      atomicAdd(&counts[1], n_v - n_ok);
    }
  }
}

// raw[lane][n_vals] bytes -> packed[pos][word]; flags lanes holding a value > 1.
// One wave = 64 witnesses x 256 positions.  Each lane reads 16 of its bytes per load (rows are
// 16-byte aligned when n_vals % 16 == 0), every byte position is turned into a 64-bit word by a
// wave ballot, and lane b keeps / stores the word of position k + b: one store per 16 positions.
// strict[k] = the mode of position k (schedule.cpp track_unreduced_values): 0xFF = a value > 1 flags the lane (it reaches
// Evaluator::get unreduced, or both a zero test and a gate); 0x01 = read by assert_zero / not alone, through copies: for the
// reference those test the unreduced INTEGER for zero (evaluator.rs:900-906,935-938), so the position is packed as
// `v != 0`; elsewhere the low bit is the residue and that is all and / xor ever look at.
// One launch packs both streams: blockIdx.z = 0 the instances, 1 the witnesses (grid.y covers the longer of the two).
struct PackArgs {
  const uint8_t* raw[2];
  const uint8_t* strict[2];
  u64* packed[2];
  u32 n_vals[2];
};
__global__ __launch_bounds__(256) void pack_inputs_kernel(const PackArgs pa, u32 batch, u32 total_words,
                                                          u32* __restrict__ lane_flags) {
  const u32 z = blockIdx.z;
  const uint8_t* __restrict__ raw = pa.raw[z];
  const uint8_t* __restrict__ strict = pa.strict[z];
  u64* __restrict__ packed = pa.packed[z];
  const u32 n_vals = pa.n_vals[z];
  if (blockIdx.y * 256 >= n_vals) return;
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
  u32 k = k0;
  if ((n_vals & 15) == 0 && ((size_t)raw & 15) == 0) {
    auto pack16 = [&](const uint4 v, u32 kk) {
      u32 w[4] = {v.x, v.y, v.z, v.w};
      const uint4 sm = *reinterpret_cast<const uint4*>(strict + kk);   // 0x00 / 0x01 / 0xFF per position (padded to 16)
      bad |= (((v.x & sm.x) | (v.y & sm.y) | (v.z & sm.z) | (v.w & sm.w)) & 0xFEFEFEFEu) != 0;   // (mode 0x01 never flags)
      const u32 m[4] = {sm.x, sm.y, sm.z, sm.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32 zm = m[q] & ~(m[q] >> 1) & 0x01010101u;   // bytes of mode 0x01 (wave-uniform: the branch is scalar)
        if (zm) {   // bit 0 of such a byte := OR of all its bits (`v != 0`)
          u32 t = w[q] | ((w[q] >> 4) & 0x0F0F0F0Fu);
          t |= (t >> 2) & 0x3F3F3F3Fu;
          t |= (t >> 1) & 0x7F7F7F7Fu;
          w[q] ^= (w[q] ^ t) & zm;
        }
      }
      u64 mine = 0;
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const u64 m = __ballot((w[b >> 2] >> (8 * (b & 3))) & 1);
        if (lane == (u32)b) mine = m;
      }
      if (lane < 16) packed[(size_t)(kk + lane) * total_words + word] = mine;
    };
    // a lane's row is its own: 64 lanes touch 64 different cache lines per load.  Four loads (64 bytes of the row) are
    // issued together so that the lines are used while they are in the L1, not fetched again per 16 bytes
    for (; k + 64 <= k1; k += 64) {
      uint4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = valid ? *reinterpret_cast<const uint4*>(row + k + 16 * q) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) pack16(v[q], k + 16 * q);
    }
    for (; k + 16 <= k1; k += 16) pack16(valid ? *reinterpret_cast<const uint4*>(row + k) : make_uint4(0, 0, 0, 0), k);
  }
  for (; k < k1; ++k) {
    const uint8_t v = valid ? row[k] : 0;
    bad |= v > 1 && strict[k] == 0xFF;
    const u64 m = __ballot(strict[k] == 0x01 ? v != 0 : (v & 1) != 0);
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
    const TapeOp op = load_op_scalar(args.ops, i);
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

// ---------------------------------------------------------------------------
// LDS-resident GF(2) replay.  After liveness analysis a Boolean relation keeps
// few wires alive at once (C4: 30,426 slots).  With 32 witnesses per 32-bit
// word that is < 160 KiB, i.e. the *whole wire table* of a 32-witness slice
// fits in one CU's LDS.  One 1024-thread workgroup owns a slice and walks the
// entire program: the ops of a level are spread over the threads, levels are
// separated by a workgroup barrier, and no wire value ever travels to HBM.
// The only streamed data is the program itself (8 bytes per op, identical for
// every workgroup, so it is served from L2).
__device__ __forceinline__ void lds_exec(const LdsOp op, u32* __restrict__ T, const BoolLdsArgs& args, u32 col,
                                         u32 valid_mask) {
  u32 r;
  switch (op.kind) {
    case OP_XOR: r = T[op.a] ^ T[op.b]; break;
    case OP_AND: r = T[op.a] & T[op.b]; break;
    case OP_NOT: r = ~T[op.a]; break;
    case OP_COPY: r = T[op.a]; break;
    case OP_CONST: r = args.consts[op.a] ? ~0u : 0u; break;
    case OP_INSTANCE: r = args.packed_inst[(size_t)(op.a | ((u32)op.b << 16)) * (2 * args.total_words64) + col]; break;
    case OP_WITNESS: r = args.packed_wit[(size_t)(op.a | ((u32)op.b << 16)) * (2 * args.total_words64) + col]; break;
    case OP_ASSERT: {
      u32 nz = T[op.a] & valid_mask;
      const u32 seq = op.dst | ((u32)op.b << 16);
      while (nz) {
        const u32 bit = __builtin_ctz(nz);
        nz &= nz - 1;
        atomicMin(&args.first_fail[col * 32 + bit], seq);
      }
      return;
    }
    default: return;
  }
  T[op.dst] = r;
}

// The and / xor / not / copy ops of a level are ONE sequence -- its `and` ops first, then the rest, which the rows know as
// `xor`: `not a` is stored as a xor ONES, a copy as a xor ZERO (lds_layout.hpp) -- cut into ROWS of 2048 ops and padded
// at its end only with ops on scratch slots: and-rows, at most one SPLIT row (its first ops `and`, its other ops `xor`),
// xor-rows.  The ops are stored without their kind: three u16 per op, two ops = 12 bytes per thread and row --
// {pair | a0 << 16, b0 | dst1 << 16, a1 | b1 << 16}.  The two
// results of a thread go to the two halves of ONE 8-byte slot pair (slots 2 * pair and 2 * pair + 1: the scheduler
// allocates the result slots of a row pair by pair), so a row step has one address shift and one ds_write_b64 for them.
// Consecutive rows of a level form BLOCKS of at most BR rows (args.hpp: block header).

// wave-uniform table read on the scalar path (s_load)
__device__ __forceinline__ u32 lds_sload(const u32* table, u32 idx) {
  typedef const u32 __attribute__((address_space(4))) cu32;
  cu32* q = (cu32*)(unsigned long long)table;
  return q[__builtin_amdgcn_readfirstlane(idx)];
}

// ---- the block pipeline --------------------------------------------------------------------------------------------
// All rows of a level are independent, so the operand reads of the next kLdsAhead rows are issued BEFORE the results of
// row r are computed and written: a wave's LDS queue holds reads and writes of neighbouring rows, and its address
// arithmetic and gate instructions run while the LDS pipe works.  Every row is exactly 4 reads + 2 writes and the LDS
// instructions of a wave complete in order, hence "row r has arrived" is `s_waitcnt lgkmcnt(#LDS instructions issued
// after its last read)`, a constant per unrolled step.  The program words arrive the same way: every block issues the
// same global loads in the same order -- the header of block k + 3, then after the writes of row r the 12 bytes of row
// r of block k + 1 into the registers row r just left, for all BR rows (the kernel's template parameter) whatever the
// blocks hold -- so "the rows this step reads are there" is always the same vmcnt: one block of program words in flight.
//
// A level ends with a barrier, and what stands next to it is the critical path of the whole CU: what a wave executes
// between its last write and the barrier delays everybody's release, what it executes between the barrier and its first
// LDS reads runs with an idle LDS.  So the code of a run of blocks is picked once per run (ldsp_run<BR, A>), a block is
// prepared -- header words, pointers, split, the addresses of its row 0 -- in the middle of the block before it (or, for
// the block sizes without spare registers, behind that block's last write), and the barrier stands at the top of the
// loop, behind the loop's own scalar book-keeping (profiles/r03_tuning_sweeps.txt has each of these measured).
//
// hipcc cannot be told that a register is waiting for a load: it is free to copy it (at a branch join, say) before the
// data is there.  So everything that is in flight lives in registers the compiler does not own -- the kernel is
// compiled for kLdsCompilerVgprs registers (amdgpu_num_vgpr) and the registers above are named in the asm text:
//   v[kRegP + 4 r .. + 2]  program words of row r: {pair | a0 << 16, b0 | dst1 << 16, a1 | b1 << 16}
//   v[kRegV + 4 s .. + 3]  operand values a0, a1, b0, b1 of the row in value set s (row r uses set r mod (kLdsAhead + 1))
//   v[kRegU .. + 3]        (block sizes <= 10) LDS addresses of the operands of the next block's row 0
//   v[kRegT .. + 3]        address / result temporaries of a row step
//   v[kRegH .. + 1]        block header in flight
// No scalar load may be in flight inside a run (they share lgkmcnt with the LDS and return out of order): block headers
// travel on the vector path.  The slot fields become LDS byte addresses by ONE SDWA shift each (the 16-bit half is the
// shifted operand; the wire table starts at LDS address 0: the kernel has no static LDS).
constexpr int kLdsCompilerVgprs = 64;
// rows whose operand reads are in flight behind the row being computed.  Fixed: two rows ahead measured the same
// (profiles/r03_tuning_sweeps.txt) and that build once computed wrong wires (the mid-block prepare only covers one row of
// program words), so it is not a switch any more -- the counted waits below are written for exactly one row.
constexpr int kLdsAhead = 1;
constexpr int kRegP = 64, kRegH = 124, kRegT = kRegH - 4;   // kRegT .. + 3: address / result temporaries of a step
template <int BR> constexpr int kRegV = kRegP + 4 * BR;   // value sets follow the BR rows of program words
template <int BR> constexpr bool kLdsFits = kRegP + 4 * BR + 4 * (kLdsAhead + 1) <= kRegT;
// Four registers between the value sets and the step temporaries, where a block size leaves them: the LDS addresses of
// the next block's row 0, computed in the middle of a block (ldsp_run)
template <int BR> constexpr int kRegU = kRegV<BR> + 4 * (kLdsAhead + 1);
template <int BR> constexpr bool kLdsEarly = kRegU<BR> + 4 <= kRegT;
static_assert((kRegT + 2) % 2 == 0, "the two results of a row step are the data of one ds_write_b64: an aligned register pair");

#define ZKGPU_SDWA_LO " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
#define ZKGPU_SDWA_HI " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"

// The kernel is bound by instruction issue (a SIMD hands out one slot every four cycles and the 16 waves of the
// workgroup fill them): a row step is two asm statements holding nothing but 7 VALU (five address shifts, two gates),
// 5 LDS (four reads, one 64-bit write), 1 VMEM and the waits (vmcnt every other row, lgkmcnt every row).  What hipcc makes of a C++ if-chain over the
// row kind is a dozen scalar instructions and up to five branches per row, and between statements it puts address adds.
template <int BR, int R, int T0 = kRegT>   // the LDS addresses of the four operands of row R, into v[T0 .. T0 + 3]
__device__ __forceinline__ void ldsp_addr() {
  constexpr int P = kRegP + 4 * R;
  asm volatile("v_lshlrev_b32_sdwa v[%[t0]], %[two], v[%[ax]]" ZKGPU_SDWA_HI
               "v_lshlrev_b32_sdwa v[%[t1]], %[two], v[%[az]]" ZKGPU_SDWA_LO
               "v_lshlrev_b32_sdwa v[%[t2]], %[two], v[%[ay]]" ZKGPU_SDWA_LO
               "v_lshlrev_b32_sdwa v[%[t3]], %[two], v[%[az]]" ZKGPU_SDWA_HI
               :
               : [two] "s"(2u), [ax] "n"(P), [ay] "n"(P + 1), [az] "n"(P + 2), [t0] "n"(T0), [t1] "n"(T0 + 1), [t2] "n"(T0 + 2),
                 [t3] "n"(T0 + 3)
               : "memory");
}
template <int BR, int R, int T0 = kRegT>   // ... and the four reads (block start: rows 0 .. kLdsAhead - 1)
__device__ __forceinline__ void ldsp_issue() {
  constexpr int V = kRegV<BR> + 4 * (R % (kLdsAhead + 1));
  asm volatile("ds_read_b32 v[%[a0]], v[%[t0]]\n\t"
               "ds_read_b32 v[%[a1]], v[%[t1]]\n\t"
               "ds_read_b32 v[%[a2]], v[%[t2]]\n\t"
               "ds_read_b32 v[%[a3]], v[%[t3]]"
               :
               : [a0] "n"(V), [a1] "n"(V + 1), [a2] "n"(V + 2), [a3] "n"(V + 3), [t0] "n"(T0), [t1] "n"(T0 + 1), [t2] "n"(T0 + 2),
                 [t3] "n"(T0 + 3)
               : "memory");
}
template <int BR>
__device__ __forceinline__ void ldsp_wait_row() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(BR - kLdsAhead) : "memory");
}
// A statement that DEFINES a register (an output, a clobbered flag) makes hipcc put an s_nop behind it on gfx940+ (it
// cannot see inside and assumes a forwarding hazard): the steps define nothing -- their four temporaries are registers
// of the hand-managed range (kRegT) -- except the one form that decides and / xor at run time (it clobbers scc).
#define ZKGPU_LDS_STEP_DST "v_lshlrev_b32_sdwa v[%[t0]], %[three], v[%[px]]" ZKGPU_SDWA_LO   /* byte address of the slot pair */
#define ZKGPU_LDS_STEP_AHEAD                                           \
  "v_lshlrev_b32_sdwa v[%[t0]], %[two], v[%[ax]]" ZKGPU_SDWA_HI        \
  "v_lshlrev_b32_sdwa v[%[t1]], %[two], v[%[az]]" ZKGPU_SDWA_LO        \
  "v_lshlrev_b32_sdwa v[%[t2]], %[two], v[%[ay]]" ZKGPU_SDWA_LO        \
  "v_lshlrev_b32_sdwa v[%[t3]], %[two], v[%[az]]" ZKGPU_SDWA_HI        \
  "ds_read_b32 v[%[a0]], v[%[t0]]\n\t"                                 \
  "ds_read_b32 v[%[a1]], v[%[t1]]\n\t"                                 \
  "ds_read_b32 v[%[a2]], v[%[t2]]\n\t"                                 \
  "ds_read_b32 v[%[a3]], v[%[t3]]\n\t"
#define ZKGPU_LDS_STEP_TEMPS [t0] "n"(kRegT), [t1] "n"(kRegT + 1), [t2] "n"(kRegT + 2), [t3] "n"(kRegT + 3)

// step R of a block of N rows: [reads of row R + kLdsAhead] -> wait for row R -> gates, writes -> refill
// The split row (lds_layout.hpp): its ops at row positions below `split` are `and`, the others `xor`.  Thread t executes
// the ops at positions 2 t and 2 t + 1 (`wpos`, `wpos + 1`: two VGPRs the kernel sets up once).
struct LdsSplit {
  u32 split;   // wave-uniform, 0 .. 2048
  u32 wpos;    // 2 * threadIdx.x
  u32 wpos1;   // 2 * threadIdx.x + 1
};

template <int BR, int N, int R, int KIND>
__device__ __forceinline__ void ldsp_step(const LdsSplit& sp, const u32* src_next, u32 voff) {
  constexpr int S = kLdsAhead + 1;
  constexpr int P = kRegP + 4 * R, V = kRegV<BR> + 4 * (R % S);
  constexpr int w = R < kLdsAhead ? R : kLdsAhead;                              // writes (one per row) behind the reads of row R
  constexpr int ahead = N - 1 - R < kLdsAhead ? N - 1 - R : kLdsAhead;         // rows whose reads are behind them
  constexpr int lg = 4 * ahead + w < 15 ? 4 * ahead + w : 15;                  // lgkmcnt counts to 15: beyond, wait for a little more
  if constexpr (R + kLdsAhead < N) {
    constexpr int A = R + kLdsAhead, PA = kRegP + 4 * A, VA = kRegV<BR> + 4 * (A % S);
    // program words of rows A and A + 1 have arrived: one vmcnt wait per two rows
    constexpr int vm = BR - kLdsAhead - 1;
#define ZKGPU_LDS_STEP_AHEAD_OPERANDS                                                                                      \
  :                                                                                                                        \
  : [two] "s"(2u), [three] "s"(3u), [vm] "n"(vm), [lg] "n"(lg), [ax] "n"(PA), [ay] "n"(PA + 1), [az] "n"(PA + 2), [a0] "n"(VA),              \
    [a1] "n"(VA + 1), [a2] "n"(VA + 2), [a3] "n"(VA + 3), [px] "n"(P), [py] "n"(P + 1), ZKGPU_LDS_STEP_TEMPS               \
  : "memory"
    if constexpr (R % 2 == 0)
      asm volatile("s_waitcnt vmcnt(%[vm])\n\t" ZKGPU_LDS_STEP_AHEAD ZKGPU_LDS_STEP_DST "s_waitcnt lgkmcnt(%[lg])" ZKGPU_LDS_STEP_AHEAD_OPERANDS);
    else
      asm volatile(ZKGPU_LDS_STEP_AHEAD ZKGPU_LDS_STEP_DST "s_waitcnt lgkmcnt(%[lg])" ZKGPU_LDS_STEP_AHEAD_OPERANDS);
  } else {
    asm volatile(ZKGPU_LDS_STEP_DST "s_waitcnt lgkmcnt(%[lg])"
                 :
                 : [three] "s"(3u), [lg] "n"(lg), [px] "n"(P), ZKGPU_LDS_STEP_TEMPS
                 : "memory");
  }
  // gates of row R, their writes, and the refill of the row's registers.  KIND 0 / 1: the row is known to be and / xor
  // (a full block: `A` and-rows, the split row, xor-rows -- one instantiation per A, no decision per row); KIND 2: the
  // split row -- every lane computes both gates and keeps the one of its side of the split: six more vector
  // instructions in one row of a level, no branch (the statement clobbers vcc).
#define ZKGPU_LDS_STEP_WRITE                                         \
  "ds_write_b64 v[%[t0]], v[%[t2]:%[t3]]\n\t"                        \
  "global_load_dwordx3 v[%[px]:%[pz]], %[voff], %[base]"
#define ZKGPU_LDS_STEP_GATES(OP)                                     \
  OP " v[%[t2]], v[%[v0]], v[%[v2]]\n\t"                             \
  OP " v[%[t3]], v[%[v1]], v[%[v3]]\n\t"
#define ZKGPU_LDS_STEP_TAIL_OPERANDS                                                                                 \
  [voff] "v"(voff), [base] "s"(src_next), [px] "n"(P), [pz] "n"(P + 2), [v0] "n"(V), [v1] "n"(V + 1), [v2] "n"(V + 2), \
      [v3] "n"(V + 3), ZKGPU_LDS_STEP_TEMPS
  if constexpr (KIND == 0) {
    asm volatile(ZKGPU_LDS_STEP_GATES("v_and_b32") ZKGPU_LDS_STEP_WRITE : : ZKGPU_LDS_STEP_TAIL_OPERANDS : "memory");
  } else if constexpr (KIND == 1) {
    asm volatile(ZKGPU_LDS_STEP_GATES("v_xor_b32") ZKGPU_LDS_STEP_WRITE : : ZKGPU_LDS_STEP_TAIL_OPERANDS : "memory");
  } else {
    asm volatile("v_and_b32 v[%[t2]], v[%[v0]], v[%[v2]]\n\t"
                 "v_xor_b32 v[%[t1]], v[%[v0]], v[%[v2]]\n\t"
                 "v_cmp_gt_u32 vcc, %[split], %[wpos]\n\t"               // the even op is an `and`: 2 t < split
                 "v_cndmask_b32 v[%[t2]], v[%[t1]], v[%[t2]], vcc\n\t"   // (vcc ? src1 : src0)
                 "v_and_b32 v[%[t3]], v[%[v1]], v[%[v3]]\n\t"
                 "v_xor_b32 v[%[t1]], v[%[v1]], v[%[v3]]\n\t"
                 "v_cmp_gt_u32 vcc, %[split], %[wpos1]\n\t"
                 "v_cndmask_b32 v[%[t3]], v[%[t1]], v[%[t3]], vcc\n\t"
                 ZKGPU_LDS_STEP_WRITE
                 :
                 : [split] "s"(sp.split), [wpos] "v"(sp.wpos), [wpos1] "v"(sp.wpos1), ZKGPU_LDS_STEP_TAIL_OPERANDS
                 : "memory", "vcc");
  }
}
template <int R>   // this thread's 12 bytes of a row -> the registers of row R; voff = byte offset of (thread, row R) in a block
__device__ __forceinline__ void ldsp_gload(const u32* block, u32 voff) {
  asm volatile("global_load_dwordx3 v[%2:%3], %0, %1" : : "v"(voff), "s"(block), "n"(kRegP + 4 * R), "n"(kRegP + 4 * R + 2) : "memory");
}
__device__ __forceinline__ void ldsp_gload_header(const u32* p, u32 vzero) {   // p is wave-uniform: SGPR base, zero offset
  const unsigned long long a = (unsigned long long)p;
  // (the builtin returns int: without the casts the low word would be sign-extended over the high one)
  p = (const u32*)(((unsigned long long)(u32)__builtin_amdgcn_readfirstlane((u32)(a >> 32)) << 32) |
                   (unsigned long long)(u32)__builtin_amdgcn_readfirstlane((u32)a));
  asm volatile("global_load_dwordx2 v[%2:%3], %0, %1" : : "v"(vzero), "s"(p), "n"(kRegH), "n"(kRegH + 1) : "memory");
}

// One block of N rows, straight-line (one instantiation per N: every wait is a constant).  A >= 0 (full blocks): the
// first A rows are and-rows, row A is the split row, the rest xor-rows; A < 0: `ad` and-rows, known at run time -- every
// row runs the split row's code with a split that puts all of it on one side (2048: all `and`; 0: all `xor`) unless it
// is the split row itself.
struct LdsNoHook {
  __device__ __forceinline__ void operator()() const {}
};
// HOOK: called once, behind row N - 2 of a full block -- where ldsp_run prepares the NEXT block (its row 0 was fetched by
// step 0, long ago by then), in the shadow of this block's last LDS waits
template <int BR, int N, int A, int R, class Hook>
__device__ __forceinline__ void ldsp_rows(const LdsSplit& sp, u32 ad, const u32* src_next, const u32 (&voff)[BR], Hook&& hook) {
  if constexpr (R < BR) {
    if constexpr (R < N) {
      if constexpr (A < 0) {
        LdsSplit row = sp;
        row.split = (u32)R < ad ? (u32)kLdsRowOps : ((u32)R == ad ? sp.split : 0u);
        ldsp_step<BR, N, R, 2>(row, src_next, voff[R]);
      } else {
        ldsp_step<BR, N, R, (R < A ? 0 : (R == A ? 2 : 1))>(sp, src_next, voff[R]);
        if constexpr (R == N - 2) hook();
      }
    } else {
      ldsp_gload<R>(src_next, voff[R]);
    }
    ldsp_rows<BR, N, A, R + 1>(sp, ad, src_next, voff, hook);
  }
}
template <int BR, int N, int A, int T0 = kRegT, class Hook = LdsNoHook>
__device__ __forceinline__ void ldsp_block(const u32* hdr_next3, u32 vzero, const LdsSplit& sp, u32 ad, const u32* src_next,
                                           const u32 (&voff)[BR], Hook&& hook = Hook()) {
  if constexpr (N <= BR && A <= N) {
    ldsp_issue<BR, 0, T0>();   // (the addresses of row 0 were computed in front of the barrier: ldsp_run)
    // (the header load stays behind the barrier: issued in front of it, with the rest of the prologue, the replay of C4
    // took 3 % longer -- profiles/r03_tuning_sweeps.txt)
    ldsp_gload_header(hdr_next3, vzero);
    ldsp_rows<BR, N, A, 0>(sp, ad, src_next, voff, hook);
  }
}
template <int BR, int R>
__device__ __forceinline__ void ldsp_gload_all(const u32* src, const u32 (&voff)[BR]) {
  if constexpr (R < BR) {
    ldsp_gload<R>(src, voff[R]);
    ldsp_gload_all<BR, R + 1>(src, voff);
  }
}

// A run of `run` blocks starting with block `first`.  Header = {descriptor, BYTE offset of the block in ops6}.  A >= 0:
// every block of the run is full and starts with A and-rows (lds_program.cpp starts a new run where that changes), so
// nothing is decided per block; A < 0: blocks of any row count, and-rows read from the header.
template <int BR, int A>
__device__ __forceinline__ void ldsp_run(const BoolLdsArgs& args, u32 first, u32 run, u32 vzero, LdsSplit sp, const u32 (&voff)[BR]) {
  if constexpr (A <= BR) {
    const u32* hdr = args.blocks + 2 * (size_t)first;
    const u32 last = __builtin_amdgcn_readfirstlane(run - 1);
    u32 d_cur = lds_sload(hdr, 0), f_cur = lds_sload(hdr, 1);
    u32 d_nxt = lds_sload(hdr, 2 * min(1u, last)), f_nxt = lds_sload(hdr, 2 * min(1u, last) + 1);
    const char* stream = reinterpret_cast<const char*>(args.ops6);
    ldsp_gload_header(hdr + 2 * min(2u, last), vzero);
    ldsp_gload_all<BR, 0>(reinterpret_cast<const u32*>(stream + f_cur), voff);
    // the scalar loads above have to be over: none may be in flight below (naming the values orders the loads before)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d_cur), "+s"(f_cur), "+s"(d_nxt), "+s"(f_nxt) : : "memory");
    // What a block needs before its first LDS read -- the header of the block after next out of its registers, the
    // pointers of the loads it issues, its split, the addresses of row 0 -- does not depend on the wire table, and behind
    // the barrier every wave of the CU would execute it with nothing to overlap it with.
    u32 d_n2, f_n2;
    const u32* hdr_next3;
    const u32* src_next;
    u32 barrier = 0;
    if constexpr (A >= 0 && kLdsEarly<BR>) {
      // Full blocks with four spare registers: block k + 1 is prepared in the MIDDLE of block k (behind its row BR - 2, in
      // the shadow of the LDS waits of its last rows), into variables of its own that take over when block k ends; the
      // addresses of its row 0 wait in the spare registers.  Nothing but the rotation is left between the last write of a
      // block and the barrier -- what stands there delays the wave's arrival at the barrier, i.e. the whole CU.
      constexpr int U = kRegU<BR>;
      u32 d_n3 = 0, f_n3 = 0, split_n = 0;
      const u32* hdr_n = hdr;
      const u32* src_n = reinterpret_cast<const u32*>(stream);
      ldsp_wait_row<BR>();
      asm volatile("v_readfirstlane_b32 %0, v[%2]\n\tv_readfirstlane_b32 %1, v[%3]" : "=s"(d_n2), "=s"(f_n2) : "n"(kRegH), "n"(kRegH + 1));
      hdr_next3 = hdr + 2 * (size_t)(u32)__builtin_amdgcn_readfirstlane(min(3u, last));
      src_next = reinterpret_cast<const u32*>(stream + f_nxt);
      sp.split = (d_cur >> kLdsBlockSplitShift) & 2047;
      ldsp_addr<BR, 0, U>();
      for (u32 k = 0; k < run; ++k) {
        if (barrier) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // does not drain vmcnt
        ldsp_block<BR, BR, A, U>(hdr_next3, vzero, sp, 0u, src_next, voff, [&]() {
          // block k + 1 (past the last block: the clamped header / offsets of the last one again -- never used).  The
          // header of block k + 3 (issued at the start of this block) and row 0 of block k + 1 (fetched by step 0) are
          // older than the BR - 2 youngest loads in flight.
          asm volatile("s_waitcnt vmcnt(%0)" : : "n"(BR - 2) : "memory");
          asm volatile("v_readfirstlane_b32 %0, v[%2]\n\tv_readfirstlane_b32 %1, v[%3]" : "=s"(d_n3), "=s"(f_n3) : "n"(kRegH), "n"(kRegH + 1));
          hdr_n = hdr + 2 * (size_t)(u32)__builtin_amdgcn_readfirstlane(min(k + 4, last));
          src_n = reinterpret_cast<const u32*>(stream + f_n2);
          split_n = (d_nxt >> kLdsBlockSplitShift) & 2047;
          ldsp_addr<BR, 0, U>();
        });
        barrier = (d_cur >> 4) & 1;
        d_cur = d_nxt;
        f_cur = f_nxt;
        d_nxt = d_n2;
        f_nxt = f_n2;
        d_n2 = d_n3;
        f_n2 = f_n3;
        hdr_next3 = hdr_n;
        src_next = src_n;
        sp.split = split_n;
      }
    } else {
      // Otherwise it is done for block k + 1 BEHIND the last write of block k and IN FRONT of the barrier, where a wave
      // waits for its writes anyway.
      auto prepare = [&](u32 k) {
        // rows 0 .. kLdsAhead - 1 of block k and the header of block k + 2 (issued a block ago) have arrived
        ldsp_wait_row<BR>();
        asm volatile("v_readfirstlane_b32 %0, v[%2]\n\tv_readfirstlane_b32 %1, v[%3]" : "=s"(d_n2), "=s"(f_n2) : "n"(kRegH), "n"(kRegH + 1));
        hdr_next3 = hdr + 2 * (size_t)(u32)__builtin_amdgcn_readfirstlane(min(k + 3, last));
        src_next = reinterpret_cast<const u32*>(stream + f_nxt);   // past the last block: re-reads it (never used)
        sp.split = (d_cur >> kLdsBlockSplitShift) & 2047;
        ldsp_addr<BR, 0>();
      };
      prepare(0);
      // The barrier behind block k stands at the TOP of iteration k + 1: hipcc puts the scalar book-keeping of the loop
      // (the rotation of the header words, the loop test) at the latch, which is then in front of the barrier and not
      // between the barrier and the first LDS reads of the next level.  (The last block's barrier is the one that ends
      // the run.)
      for (u32 k = 0; k < run; ++k) {
        if (barrier) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // does not drain vmcnt
        if constexpr (A >= 0) {
          ldsp_block<BR, BR, A>(hdr_next3, vzero, sp, 0u, src_next, voff);
        } else {
          const u32 ad = (d_cur >> kLdsBlockAndShift) & 15;
#define ZKGPU_LDS_ANY(N) case N: ldsp_block<BR, N, -1>(hdr_next3, vzero, sp, ad, src_next, voff); break;
          switch (d_cur & 15) {
            ZKGPU_LDS_ANY(1) ZKGPU_LDS_ANY(2) ZKGPU_LDS_ANY(3) ZKGPU_LDS_ANY(4) ZKGPU_LDS_ANY(5) ZKGPU_LDS_ANY(6)
            ZKGPU_LDS_ANY(7) ZKGPU_LDS_ANY(8) ZKGPU_LDS_ANY(9) ZKGPU_LDS_ANY(10) ZKGPU_LDS_ANY(11)
            default: ldsp_block<BR, 12, -1>(hdr_next3, vzero, sp, ad, src_next, voff); break;
          }
#undef ZKGPU_LDS_ANY
        }
        barrier = (d_cur >> 4) & 1;
        d_cur = d_nxt;
        f_cur = f_nxt;
        d_nxt = d_n2;
        f_nxt = f_n2;
        if (k + 1 < run) prepare(k + 1);
      }
    }
    (void)f_cur;
  }
}

// BR: rows every block fetches (the host picks the instantiation that fetches least for the program at hand)
template <int BR>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_num_vgpr(kLdsCompilerVgprs))) void bool_lds_kernel(const BoolLdsArgs args) {
  static_assert(BR > kLdsAhead && BR <= kLdsMaxBlockRows && kLdsAhead == 1 && kLdsFits<BR>, "block rows");
  extern __shared__ __attribute__((aligned(16))) u32 T[];
  asm volatile("" ::: "v125");   // the highest hand-managed register: the kernel is allocated 126 (-> 128) registers
  const u32 tid = threadIdx.x;
  const u32 col = blockIdx.x;
  const u32 lane0 = col * 32;
  const u32 valid_mask = lane0 >= args.batch ? 0u
                         : (args.batch - lane0 >= 32 ? ~0u : ((1u << (args.batch - lane0)) - 1));
  u32 vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));   // (opaque to hipcc: stays a register, the SGPR-base form of a load needs one)
  u32 voff[BR];   // byte offset of this thread's 12-byte record of row r within a block of the stream
#pragma unroll
  for (int r = 0; r < BR; ++r) voff[r] = 12 * tid + r * (12 * 1024);
  if (tid == 0) {
    T[args.n_slots - kLdsExtraSlots + kLdsZeroSlot] = 0u;
    T[args.n_slots - kLdsExtraSlots + kLdsOnesSlot] = ~0u;
  }
  __syncthreads();
  u32 c = 0;
  while (c < args.n_chunks) {
    c = __builtin_amdgcn_readfirstlane(c);
    const u32 first = lds_sload(args.chunks, 4 * c), rows = lds_sload(args.chunks, 4 * c + 1),
              flags = lds_sload(args.chunks, 4 * c + 2), run = lds_sload(args.chunks, 4 * c + 3);
    ++c;
    if (!((flags >> 10) & 1)) {
      // generic chunk: inputs, constants, asserts, NOP padding, or a sequential (narrow-level) segment
      if (flags & kLdsChunkWave) {
        // a run of narrow levels: wave 0 walks its packets of 64 entries, lane l executing entry l (lds_layout.hpp).  The
        // entries of eight packets are fetched while the eight before them run: the program stream is the only thing that
        // comes from memory, and a wave's LDS accesses complete in order, so nothing else is waited for between levels.
        if (tid < (u32)kLdsPacketOps) {
          // (entries as the 8-byte words they are: {dst, a, b, kind} little-endian; 0 = a no-op)
          const u64* __restrict__ w = reinterpret_cast<const u64*>(args.ops) + first + tid;
          auto entry = [](u64 x) { return LdsOp{(unsigned short)x, (unsigned short)(x >> 16), (unsigned short)(x >> 32), (unsigned short)(x >> 48)}; };
          // one packet: every lane an and / xor (not and copy are xor with ONES / ZERO, padding is ZERO xor ZERO into a scratch
          // slot: lds_program.cpp) -- two reads, one select, one write, no branch; the few lanes that hold an input, a
          // constant or an assert run theirs afterwards
          auto packet = [&](u64 x) {
            const u32 kind = (u32)(x >> 48), dst = (u32)x & 0xFFFFu, a = (u32)(x >> 16) & 0xFFFFu, b = (u32)(x >> 32) & 0xFFFFu;
            const bool gate = kind == OP_AND || kind == OP_XOR;
            const u32 va = T[gate ? a : 0u], vb = T[gate ? b : 0u];
            if (gate) T[dst] = kind == OP_AND ? (va & vb) : (va ^ vb);
            if (__ballot(!gate && kind != OP_NOP) != 0ull) {
              if (!gate) lds_exec(entry(x), T, args, col, valid_mask);
            }
          };
          // eight packets in hand, eight on their way (the compiler owns 64 registers here): the program stream is the only
          // thing that comes from memory
          u64 c[8], n[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) c[q] = (u32)q < rows ? w[(size_t)q * kLdsPacketOps] : 0ull;
          for (u32 p0 = 0; p0 < rows; p0 += 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) n[q] = p0 + 8 + q < rows ? w[(size_t)(p0 + 8 + q) * kLdsPacketOps] : 0ull;
#pragma unroll
            for (int q = 0; q < 8; ++q) packet(c[q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) c[q] = n[q];
          }
        }
      } else if ((flags >> 9) & 1) {
        if (tid == 0)
          for (u32 i = 0; i < rows; ++i) lds_exec(args.ops[first + i], T, args, col, valid_mask);
      } else if ((flags & 0xFF) == OP_INSTANCE || (flags & 0xFF) == OP_WITNESS) {
        // input loads, eight at a time: entries, then the packed words (a 4-byte gather each), then the LDS stores --
        // one after the other every op paid two global latencies (16 ops per thread at C4: 25 us of a 0.93 ms replay)
        const u32 kind = flags & 0xFF;
        const u32* __restrict__ src = kind == OP_INSTANCE ? args.packed_inst : args.packed_wit;
        for (u32 j0 = 0; j0 < 2 * rows; j0 += 8) {
          LdsOp o[8];
          u32 v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            o[q] = LdsOp{0, 0, 0, (unsigned short)OP_NOP};
            if (j0 + q < 2 * rows) o[q] = args.ops[first + (j0 + q) * 1024 + tid];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q)
            v[q] = o[q].kind == kind ? src[(size_t)(o[q].a | ((u32)o[q].b << 16)) * (2 * args.total_words64) + col] : 0u;
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (o[q].kind == kind) T[o[q].dst] = v[q];
        }
      } else {
        for (u32 j = 0; j < 2 * rows; ++j) lds_exec(args.ops[first + j * 1024 + tid], T, args, col, valid_mask);
      }
      if ((flags >> 8) & 1) __syncthreads();
      continue;
    }
    // A run of `run` blocks starting with block `first`: all full blocks with the same number of and-rows (flags bits
    // 11..14; the code of such a run knows every row's kind but the split row's), or blocks of any shape (15)
#define ZKGPU_LDS_RUN(A) case A: ldsp_run<BR, A>(args, first, run, vzero, sp, voff); break;
    LdsSplit sp;
    sp.split = 0;
    sp.wpos = 2 * tid;
    sp.wpos1 = 2 * tid + 1;
    switch ((flags >> kLdsChunkAndShift) & 15) {
      ZKGPU_LDS_RUN(0) ZKGPU_LDS_RUN(1) ZKGPU_LDS_RUN(2) ZKGPU_LDS_RUN(3) ZKGPU_LDS_RUN(4) ZKGPU_LDS_RUN(5) ZKGPU_LDS_RUN(6)
      ZKGPU_LDS_RUN(7) ZKGPU_LDS_RUN(8) ZKGPU_LDS_RUN(9) ZKGPU_LDS_RUN(10) ZKGPU_LDS_RUN(11) ZKGPU_LDS_RUN(12)
      default: ldsp_run<BR, -1>(args, first, run, vzero, sp, voff); break;
    }
#undef ZKGPU_LDS_RUN
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (args.writeback) {
    // table[lane_block][slot][word64] as u32 halves: col -> (lane block, word64, half)
    const u32 lb = col / 128, w64 = (col % 128) / 2, half = col % 2;
    u32* out = reinterpret_cast<u32*>(args.table);
    const u32 n_real = args.n_slots - kLdsExtraSlots;   // without the scratch / constant slots
    for (u32 s = tid; s < n_real; s += 1024)
      out[(((size_t)lb * n_real + s) * 64 + w64) * 2 + half] = T[s];
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
