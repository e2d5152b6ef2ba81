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
// strict[k] != 0: a value > 1 at position k flags the lane (it would reach assert_zero / not unreduced, see
// schedule.cpp mark_strict_sources); elsewhere the low bit is the residue and that is all and / xor ever look at.
__global__ __launch_bounds__(256) void pack_inputs_kernel(const uint8_t* __restrict__ raw, u32 n_vals, u32 batch,
                                                          u32 total_words, u64* __restrict__ packed,
                                                          u32* __restrict__ lane_flags, const uint8_t* __restrict__ strict) {
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
    for (; k + 16 <= k1; k += 16) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (valid) v = *reinterpret_cast<const uint4*>(row + k);
      const u32 w[4] = {v.x, v.y, v.z, v.w};
      const uint4 sm = *reinterpret_cast<const uint4*>(strict + k);   // 0x00 / 0xFF per position (padded to 16)
      bad |= (((v.x & sm.x) | (v.y & sm.y) | (v.z & sm.z) | (v.w & sm.w)) & 0xFEFEFEFEu) != 0;
      u64 mine = 0;
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const u64 m = __ballot((w[b >> 2] >> (8 * (b & 3))) & 1);
        if (lane == (u32)b) mine = m;
      }
      if (lane < 16) packed[(size_t)(k + lane) * total_words + word] = mine;
    }
  }
  for (; k < k1; ++k) {
    const uint8_t v = valid ? row[k] : 0;
    bad |= v > 1 && strict[k];
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

// All ops of a chunk belong to one level, hence are mutually independent: every operand read is
// issued before the first result is written, so LDS latency is paid once per chunk, not per op.
// The ops of the kind-uniform chunks are stored without their kind (it is in the chunk header): three u16 per op,
// two ops = 12 bytes per thread and row -- {dst0 | a0 << 16, b0 | dst1 << 16, a1 | b1 << 16} -- because the program
// stream, which every workgroup reads in full, is what bounds this kernel once the LDS accesses are conflict-free.
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));  // native vector: usable as an asm operand

#ifndef ZKGPU_LDS_DIAG_BITS
#define ZKGPU_LDS_DIAG_BITS 0  // 1 no level barriers, 2 no LDS writes, 4 no LDS reads, 8 no program fetch: wrong results
#endif

// LDS byte address of the slot in the low / high half of a program word: (half << 2) in ONE instruction -- SDWA
// selects the 16-bit half as the shifted operand (hipcc emits v_and / v_bfe + v_lshl_add: two per field, twelve per
// pair of ops, most of this kernel's VALU work).  The wire table starts at LDS address 0 (the kernel has no static LDS).
__device__ __forceinline__ u32 lds_addr_lo(u32 word) {
  u32 r;
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "s"(2u), "v"(word));
  return r;
}
__device__ __forceinline__ u32 lds_addr_hi(u32 word) {
  u32 r;
  asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "s"(2u), "v"(word));
  return r;
}
typedef u32 __attribute__((address_space(3))) lds_u32;
__device__ __forceinline__ u32 lds_get(u32 byte_addr) { return *(lds_u32*)(uintptr_t)byte_addr; }
__device__ __forceinline__ void lds_put(u32 byte_addr, u32 v) { *(lds_u32*)(uintptr_t)byte_addr = v; }

// dbg: timing diagnostics only (results are wrong with any bit set): 1 no barriers, 2 no LDS writes, 4 no LDS reads
template <u32 KIND, bool FULL>
__device__ __forceinline__ void lds_rows(const u32x3 (&raw)[kLdsRows], u32 rows, u32* __restrict__ T, u32 dbg = 0) {
  (void)T;
  u32 x[2 * kLdsRows], y[2 * kLdsRows];
#pragma unroll
  for (int j = 0; j < kLdsRows; ++j) {
    if (FULL || (u32)j < rows) {
      if (dbg & 4) {
        x[2 * j] = raw[j].x; x[2 * j + 1] = raw[j].z; y[2 * j] = raw[j].y; y[2 * j + 1] = raw[j].z;
        continue;
      }
      x[2 * j] = lds_get(lds_addr_hi(raw[j].x));
      x[2 * j + 1] = lds_get(lds_addr_lo(raw[j].z));
      if (KIND == OP_XOR || KIND == OP_AND) {
        y[2 * j] = lds_get(lds_addr_lo(raw[j].y));
        y[2 * j + 1] = lds_get(lds_addr_hi(raw[j].z));
      }
    }
  }
#pragma unroll
  for (int j = 0; j < kLdsRows; ++j) {
    if (FULL || (u32)j < rows) {
      u32 r0, r1;
      if (KIND == OP_XOR) { r0 = x[2 * j] ^ y[2 * j]; r1 = x[2 * j + 1] ^ y[2 * j + 1]; }
      else if (KIND == OP_AND) { r0 = x[2 * j] & y[2 * j]; r1 = x[2 * j + 1] & y[2 * j + 1]; }
      else if (KIND == OP_NOT) { r0 = ~x[2 * j]; r1 = ~x[2 * j + 1]; }
      else { r0 = x[2 * j]; r1 = x[2 * j + 1]; }
      if (dbg & 2) {
        asm volatile("" ::"v"(r0), "v"(r1));
        continue;
      }
      lds_put(lds_addr_lo(raw[j].x), r0);
      lds_put(lds_addr_hi(raw[j].y), r1);
    }
  }
}

// wave-uniform table read on the scalar path (s_load): a vector load here would sit in vmcnt and
// force the program prefetch to drain
__device__ __forceinline__ u32 lds_sload(const u32* table, u32 idx) {
  typedef const u32 __attribute__((address_space(4))) cu32;
  cu32* q = (cu32*)(unsigned long long)table;
  return q[__builtin_amdgcn_readfirstlane(idx)];
}

// workgroup barrier for LDS data only: waits for this wave's LDS traffic, not for the global loads of
// the program prefetch (a plain __syncthreads() would drain them: vmcnt(0))
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Program-stream loads hidden from hipcc's s_waitcnt bookkeeping (cdna_hip_programming.md 5.7): the
// compiler would drain vmcnt(0) at every join of the kind/rows branches; here the waits are counted by
// hand.  The destination registers are only consumed behind lds_wait_vm<N>, which names them "+v".
__device__ __forceinline__ void lds_gload12(u32x3& dst, const u32* p) {
  asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_vm(u32x3 (&b)[kLdsRows]) {
  static_assert(kLdsRows == 4, "operand list below names 4 rows");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void lds_simple_chunk(const u32x3 (&buf)[kLdsRows], u32 rows, u32 flags,
                                                 u32* __restrict__ T, u32 dbg = 0) {
  const u32 kind = flags & 0xFF;
  if (rows == kLdsRows) {
    if (kind == OP_XOR) lds_rows<OP_XOR, true>(buf, rows, T, dbg);
    else if (kind == OP_AND) lds_rows<OP_AND, true>(buf, rows, T, dbg);
    else if (kind == OP_NOT) lds_rows<OP_NOT, true>(buf, rows, T, dbg);
    else lds_rows<OP_COPY, true>(buf, rows, T, dbg);
  } else {
    if (kind == OP_XOR) lds_rows<OP_XOR, false>(buf, rows, T, dbg);
    else if (kind == OP_AND) lds_rows<OP_AND, false>(buf, rows, T, dbg);
    else if (kind == OP_NOT) lds_rows<OP_NOT, false>(buf, rows, T, dbg);
    else lds_rows<OP_COPY, false>(buf, rows, T, dbg);
  }
  if (((flags >> 8) & 1) && !(dbg & 1)) lds_barrier();
}

__global__ __launch_bounds__(1024) void bool_lds_kernel(const BoolLdsArgs args) {
  extern __shared__ __attribute__((aligned(16))) u32 T[];
  const u32 tid = threadIdx.x;
  const u32 col = blockIdx.x;
  const u32 lane0 = col * 32;
  const u32 valid_mask = lane0 >= args.batch ? 0u
                         : (args.batch - lane0 >= 32 ? ~0u : ((1u << (args.batch - lane0)) - 1));
  const u32* __restrict__ prog = args.ops6 + 3 * tid;   // this thread's 12-byte record of a row
  // timing experiments (tools/build_variant.sh + tools/c4_diag.py): a constant, so that the switches cost nothing
  constexpr u32 dbg = ZKGPU_LDS_DIAG_BITS;
  u32 c = 0;
  while (c < args.n_chunks) {
    c = __builtin_amdgcn_readfirstlane(c);
    const u32 first = lds_sload(args.chunks, 4 * c), rows = lds_sload(args.chunks, 4 * c + 1),
              flags = lds_sload(args.chunks, 4 * c + 2), run = lds_sload(args.chunks, 4 * c + 3);
    if (run == 0) {
      // generic chunk: inputs, constants, asserts, NOP padding, or a sequential (narrow-level) segment
      if ((flags >> 9) & 1) {
        if (tid == 0)
          for (u32 i = 0; i < rows; ++i) lds_exec(args.ops[first + i], T, args, col, valid_mask);
      } else {
        for (u32 j = 0; j < 2 * rows; ++j) lds_exec(args.ops[first + j * 1024 + tid], T, args, col, valid_mask);
      }
      if ((flags >> 8) & 1) __syncthreads();
      ++c;
      continue;
    }
    // A run of `run` simple chunks (xor / and / not / copy).  The program stream is the only global traffic; three
    // chunks (3 x 48 KiB per CU) stay in flight.  Fetches are unconditional (index clamped to the run), the body holds
    // no other vector-memory op and the barrier does not drain vmcnt, so the vmcnt waits are counted by hand.
    const u32 e = c + run, c_start = c;
    u32x3 b0[kLdsRows], b1[kLdsRows], b2[kLdsRows];
    // Chunk headers {first record, rows, kind | flags, -} ride in SGPRs three chunks ahead of their use, like the
    // program words.  Each is read right AFTER a fetch has used its predecessor and right BEFORE a chunk's LDS work:
    // its latency is covered by the lgkmcnt wait the LDS reads need anyway (scalar loads and LDS share that counter;
    // read at the point of use, a header cost every wave a scalar-cache round trip twice per chunk: 1.34 -> 1.27 ms).
    // (Carrying the headers on the vector path instead -- an 8-byte load per chunk, v_readfirstlane -- with every chunk
    // owning kLdsRows rows of the stream so that fetch addresses need no header was measured slower: 1.49 ms; the
    // stream grows from 63 to 93 MB and its first touch per XCD is an L2 miss.)
    struct Hdr {
      u32 first, rows, flags;
    };
    auto header = [&](u32 cc) {
      typedef const u32 __attribute__((address_space(4))) cu32;
      cu32* q = (cu32*)(unsigned long long)(args.chunks + 4 * (size_t)__builtin_amdgcn_readfirstlane(min(cc, e - 1)));
      Hdr h;
      h.first = q[0];
      h.rows = q[1];
      h.flags = q[2];
      return h;
    };
    auto fetch = [&](u32x3 (&buf)[kLdsRows], const Hdr& h) {
      if ((dbg & 8) && c != c_start) return;
      const u32* src = prog + 3 * (size_t)h.first;   // chunk field 0: first thread record
#pragma unroll
      for (int j = 0; j < kLdsRows; ++j) lds_gload12(buf[j], src + j * (3 * 1024));
    };
    // in flight at every wait: the chunk about to run + the two behind it = 12 loads -> vmcnt(8)
    Hdr h0 = header(c), h1 = header(c + 1), h2 = header(c + 2);
    fetch(b0, h0);
    fetch(b1, h1);
    for (; c < e; c += 3) {
      fetch(b2, h2);
      __builtin_amdgcn_sched_barrier(0);
      const Hdr h3 = header(c + 3);
      lds_wait_vm<2 * kLdsRows>(b0);
      lds_simple_chunk(b0, h0.rows, h0.flags, T, dbg);
      fetch(b0, h3);
      __builtin_amdgcn_sched_barrier(0);
      const Hdr h4 = header(c + 4);
      lds_wait_vm<2 * kLdsRows>(b1);
      if (c + 1 < e) lds_simple_chunk(b1, h1.rows, h1.flags, T, dbg);
      fetch(b1, h4);
      __builtin_amdgcn_sched_barrier(0);
      const Hdr h5 = header(c + 5);
      lds_wait_vm<2 * kLdsRows>(b2);
      if (c + 2 < e) lds_simple_chunk(b2, h2.rows, h2.flags, T, dbg);
      h0 = h3;
      h1 = h4;
      h2 = h5;
    }
    c = e;
    // drain the clamped over-fetches before their registers can be reused
    lds_wait_vm<0>(b0);
    lds_wait_vm<0>(b1);
    lds_wait_vm<0>(b2);
    __syncthreads();
  }
  if (args.writeback) {
    // table[lane_block][slot][word64] as u32 halves: col -> (lane block, word64, half)
    const u32 lb = col / 128, w64 = (col % 128) / 2, half = col % 2;
    u32* out = reinterpret_cast<u32*>(args.table);
    const u32 n_real = args.n_slots - 32;   // without the scratch slots of the padding ops
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
