// Layout of the program of the LDS-resident GF(2) kernel (device/bool_kernels.hpp), shared by the host code that
// writes it (lds_program.cpp, plain C++) and the kernel that reads it.  No HIP types here.
#pragma once
#include <stdint.h>

namespace zkgpu {

struct LdsOp {  // 8-byte program entry of a generic chunk; wide fields are split over the halves the kind leaves unused
  unsigned short dst, a, b, kind;
};

// chunk = {first, rows, kind | barrier_after << 8 | sequential << 9 | blocks << 10, run}.  A generic chunk (inputs,
// constants, asserts, sequential segments) holds 8-byte entries: `first` indexes `ops`, `rows` counts rows of
// kLdsRowOps entries (or entries, for a sequential chunk).  A chunk with bit 10 set is a run of `run` BLOCKS starting at
// block `first`: the xor / and / not / copy ops of a level as rows of kLdsRowOps ops of one kind, 12 bytes per thread
// and row in `ops6` (two ops of three u16 {dst, a, b}; the dst field of the EVEN op holds the slot PAIR, dst / 2: the two
// results of a thread are the halves of one aligned 8-byte pair of slots, written by one ds_write_b64).  Block header (two u32 in `blocks`):
//   { rows (1..block_rows) | barrier_after << 4 | (row r is xor) << (kLdsBlockKindShift + r) | (a + 1) << kLdsBlockAndShift
//     when the block is full and its rows are `a` and-rows followed by xor-rows (0 otherwise),  byte offset of the block's first row in ops6 }
// Rows know two kinds only, and / xor: `not a` is stored as a xor ONES and a copy as a xor ZERO, two constant slots
// behind the kLdsScratchSlots scratch slots of the padding ops (the table holds n_slots + kLdsExtraSlots words).
constexpr int kLdsRowOps = 2048;
constexpr int kLdsMaxBlockRows = 12;   // block_rows: 4, 6, 8, 9, 10 or 12 (one kernel instantiation each)
constexpr int kLdsBlockKindShift = 5, kLdsBlockAndShift = 17;
constexpr uint32_t kLdsScratchSlots = 32, kLdsZeroSlot = 32, kLdsOnesSlot = 33, kLdsExtraSlots = 34;   // offsets past the real slots
constexpr uint32_t kLdsChunkBarrier = 1u << 8, kLdsChunkSequential = 1u << 9, kLdsChunkBlocks = 1u << 10;

}  // namespace zkgpu
