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
// block `first`: the and / xor / not / copy ops of a level as ONE sequence -- first its `and` ops, then everything else,
// which the rows know as `xor` (`not a` is stored as a xor ONES, a copy as a xor ZERO: two constant slots behind the
// kLdsScratchSlots scratch slots of the padding ops; the table holds n_slots + kLdsExtraSlots words) -- cut into rows of
// kLdsRowOps ops and padded at its END only: 12 bytes per thread and row in `ops6` (two ops of three u16 {dst, a, b}; the
// dst field of the EVEN op holds the slot PAIR, dst / 2: the two results of a thread are the halves of one aligned 8-byte
// pair of slots, written by one ds_write_b64).  So the rows of a level are and-rows, then at most one SPLIT row -- its
// first ops `and`, its other ops `xor` -- then xor-rows.  Block header (two u32 in `blocks`):
//   { rows (1..block_rows) | barrier_after << 4 | A << kLdsBlockAndShift | split << kLdsBlockSplitShift,
//     byte offset of the block's first row in ops6 }
// A = and-rows at the start of the block; the first `split` ops (0 .. 2047) of row A of the block (if it has one) are
// `and`, its other ops `xor` (thread t of the workgroup executes ops 2 t and 2 t + 1 of a row), and the rows behind it are
// xor-rows.  A run of blocks (one chunk) holds either full blocks with one and the same A -- chunk flags bits
// kLdsChunkAndShift.. = A, the kernel's code for the run knows every row's kind but the split row's -- or blocks of any
// shape (15).
constexpr int kLdsRowOps = 2048;
constexpr int kLdsMaxBlockRows = 12;   // block_rows: 4, 6, 8, 9, 10 or 12 (one kernel instantiation each)
constexpr int kLdsBlockAndShift = 5, kLdsBlockSplitShift = 9;   // 4 and 11 bits
constexpr int kLdsChunkAndShift = 11;                            // 4 bits of the chunk's flags word
constexpr uint32_t kLdsScratchSlots = 32, kLdsZeroSlot = 32, kLdsOnesSlot = 33, kLdsExtraSlots = 34;   // offsets past the real slots
constexpr uint32_t kLdsChunkBarrier = 1u << 8, kLdsChunkSequential = 1u << 9, kLdsChunkBlocks = 1u << 10;
// A sequential chunk with kLdsChunkWave: a run of NARROW levels (fewer than ScheduleOptions::bool_narrow_width ops each)
// as `rows` PACKETS of kLdsPacketOps entries, every level padded to whole packets with no-ops.  Wave 0 of the workgroup walks
// the packets in order, lane l executing entry l of a packet: the entries of a packet are independent (one level), and a
// wave's LDS accesses complete in order, so a level's writes are seen by the next level's reads without any barrier -- one
// workgroup barrier for the whole run instead of one per level and a padded 2048-op row each.
constexpr uint32_t kLdsChunkWave = 1u << 15;
constexpr int kLdsPacketOps = 64;

}  // namespace zkgpu
