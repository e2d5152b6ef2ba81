// Tape -> device program.  The reference executes backend calls one at a time
// (rust/src/consumers/evaluator.rs:288-301); the calls only depend on each
// other through wires, so the tape is re-ordered into dependency levels
// (every op of a level is independent -> one wide kernel launch), operand
// handles are renamed to wire-table slots with liveness-based reuse, and runs
// of very narrow levels are fused into sequential launches walked by one
// wavefront per lane block.
//
// The scheduler works on WINDOWS of the tape (StreamScheduler): the reference consumes a relation as a stream of
// <= 100k-gate messages (evaluator.rs:286-301, producers/builder.rs:46-74), and a window of the tape can be scheduled --
// and its part of the program sent to the GPU -- while later messages are still being parsed.  What a window needs to
// know about the future is which of its values can still get readers: that is what the drop records of the tape say
// (tape.hpp).  build_schedule() is the one-window case (the whole tape, everything but the pinned wires closed).
#pragma once
#include <stdint.h>
#include <utility>
#include <vector>

#include "tape.hpp"

namespace zki {

// A slot number with this bit names a value of a STRAND that never leaves the strand (produced in it, last read in it, nobody
// can ask for it afterwards): it lives in the LDS of the workgroup that walks the strand, [k][chunk][lane], not in the wire
// table (device/args.hpp kSlotInLds).  Only entries of a strand's launch may carry such slots.
constexpr uint32_t kSlotInLds = 0x40000000u;
constexpr uint32_t kStrandLdsBytes = 128 * 1024;   // of the CU's 160 KiB

struct DevOp {  // == zkgpu::TapeOp (device/replay_kernels.hpp)
  uint32_t dst, a, b, kind;
};

// Program entry when gate fusion is on: an Add/Mul operand may itself be an Add/Mul of two slots that is
// evaluated in registers (the absorbed producer is never written to the wire table).
//   kind: bits 0-7 main op (TapeKind), bits 8-9 / 10-11: operand a / b is 0 = slot a0 / b0,
//         1 = add(a0,a1), 2 = mul(a0,a1)
struct DevOp2 {  // == zkgpu::TapeOp2 (device/replay_kernels.hpp)
  uint32_t dst, kind, a0, a1, b0, b1, pad0, pad1;
};

struct Launch {
  uint32_t first = 0;         // index into Schedule::ops
  uint32_t count = 0;
  uint32_t ops_per_wave = 1;  // count => one wave walks the whole launch in order
  bool sequential = false;    // ops depend on each other: no operand prefetch
  uint32_t level_begin = 0, level_end = 0;
  uint32_t level_ptr = 0;     // sequential launch of the fused format (a strand): index into Schedule::strand_level_ptr of
                              // its level_end - level_begin + 1 entry offsets (relative to `first`)
  uint32_t window = 0;        // tape window the launch belongs to (its entries were uploaded with that window)
  uint32_t hot_count = 0;     // a level of the fused program: its first hot_count entries are the Add/Mul ones
  uint32_t strand_levels = 0; // a strand: its non-empty levels = the intervals of its level_ptr list (level_ptr holds one more entry)
  uint32_t lds_slots = 0;     // a strand: values that live in the workgroup's LDS instead of the wire table (slots kSlotInLds | k, k < lds_slots)
  bool has_bitops = false;    // holds and / xor over an odd field (the kernels' cold instantiation)
};

struct ScheduleOptions {
  bool retain_all = false;          // every value keeps its own slot (wire dumps for parity tests)
  uint32_t narrow_width = 3;        // levels with fewer ops than this are fused into sequential launches
  uint32_t bool_narrow_width = 257; // ... GF(2): levels with fewer ops than this (at most four packets of 64) do not run as padded
                                    // 2048-op rows with a workgroup barrier each: consecutive ones form a run that ONE wave
                                    // of the LDS-resident kernel walks packet by packet (lds_layout.hpp kLdsChunkWave)
  uint32_t strand_width = 17;       // ... with the fused entry format: into strands (one workgroup per lane block walks the
                                    // levels with a barrier between them, device/replay_kernels.hpp replay_strand_kernel)
  int sort_by_operand = 3;          // order of a level's ops: 0 tape order, 1 by first-operand slot, 2 that + shared-operand walk, 3 the walk alone
  bool fuse = true;                 // absorb single-reader Add/Mul producers into their consumer (never with retain_all)
  bool fermat = true;               // a Switch exponent ladder x^(p-1), p prime, becomes one `x != 0` entry (never with retain_all)
  bool pair = true;                 // one entry for the two same-level readers of a producer nobody else reads (never with retain_all)
  bool propagate_copies = true;     // readers use a copy's source; unobserved copies are not materialised (never with retain_all)
  std::vector<uint32_t> pinned;     // handles that must stay readable after the replay (Evaluator::get)
  bool pinned_are_carried = false;  // ... because the next field segment takes them over from the wire table (capi.cpp)
  uint32_t threads = 0;             // worker threads for the per-level ordering (0 = min(8, hardware threads))
  bool bank_aware = true;           // GF(2): order the ops and number the slots so that one LDS instruction hits 32 banks
  bool strand_lds = true;           // strands keep the values that never leave them in LDS (kSlotInLds)
  bool strand_prefetch = true;      // ... and copy what they read out of the wire table into LDS a few levels ahead (needs strand_lds)
  bool strand_merge = true;         // ... and levels that need no barrier between them are one level (needs strand_lds)
  bool strand_reassociate = true;   // ... and (z * x) * y with z fresh off the chain and x, y long there keeps z * (x * y) (needs strand_lds)
  bool strand_split_inputs = true;  // ... and an input's fetch runs a level or more ahead of its conversion (needs strand_lds)
};

struct Schedule {
  std::vector<DevOp> ops;
  std::vector<DevOp2> ops2;         // used instead of `ops` when fused
  bool fused = false;
  uint64_t n_absorbed = 0;
  uint64_t n_copies_elided = 0;
  uint64_t n_ladders = 0;           // exponent ladders replaced by one entry each
  uint64_t n_paired = 0;            // producers evaluated inside a pair entry (counted in n_absorbed too)
  uint64_t n_strand_inputs_split = 0;    // inputs of a strand fetched ahead of their conversion
  uint64_t n_strand_reassociated = 0;    // products of a strand computed off its dependency chain
  uint64_t n_strand_levels_joined = 0;   // strand levels that run behind the level in front of them without a barrier
  uint64_t n_strand_prefetches = 0; // copy entries that bring a strand's wire-table operands into LDS ahead of their reader
  std::vector<Launch> launches;
  std::vector<uint32_t> slot_of;    // per tape op: slot of its value (kNoWire for asserts)
  std::vector<uint32_t> level_of;   // per tape op
  // input positions whose value must be canonical (an unreduced value would reach copy / assert_zero / not / a bit
  // operation / Evaluator::get): the GF(2) input packing flags a lane only for these (arithmetic entries carry the flag)
  std::vector<uint8_t> strict_instance, strict_witness;
  std::vector<uint8_t> strict_carry;       // the same per value carried in from the previous field segment (TK_CARRY)
  // Wires alive at the end that are an input the relation has only copied: Evaluator::get returns the unreduced integer
  // (evaluator.rs:750-752), so zkgpu_get_wire reads the input, not the wire table.  (tape handle, 2 + 4 * position + stream),
  // sorted by handle.
  std::vector<std::pair<uint32_t, uint32_t>> raw_source;
  // Constants >= p whose unreduced bits are read (and / xor over a field other than GF(2), Evaluator::get): the tape's
  // constant indices, in the order of the RAW entries the constant pool holds behind its device-form entries
  // (const_words: n_consts device-form constants, then these as plain little-endian integers of words_per_const words);
  // stream 3 of the source codes names them by their position here.
  std::vector<uint32_t> raw_const_of;
  std::vector<uint32_t> strand_level_ptr;  // level bounds of the strands (see Launch::level_ptr)
  std::vector<uint64_t> window_first_op;   // per window: index of its first program entry (+ a final end marker)
  uint32_t n_slots = 0;
  uint32_t n_levels = 0;
  uint32_t max_level_width = 0;
  bool retain_all = false;
  bool boolean_path = false;        // p == 2: bit-packed wires
  bool has_bitops = false;          // some entry is and / xor over an odd field
  // constant pool in device form: arithmetic = Montgomery words (nwords each);
  // boolean = one u32 (0/1) per constant
  std::vector<uint32_t> const_words;
  uint32_t words_per_const = 0;

  // everything but the per-entry and per-handle arrays (ops, ops2, slot_of, level_of): what the engine keeps once the
  // program is on the device (a 10 M-gate program is a quarter of a gigabyte of those)
  Schedule without_entries() const {
    Schedule c;
    c.fused = fused;
    c.n_absorbed = n_absorbed;
    c.n_copies_elided = n_copies_elided;
    c.n_ladders = n_ladders;
    c.n_paired = n_paired;
    c.n_strand_prefetches = n_strand_prefetches;
    c.n_strand_levels_joined = n_strand_levels_joined;
    c.n_strand_reassociated = n_strand_reassociated;
    c.n_strand_inputs_split = n_strand_inputs_split;
    c.launches = launches;
    c.strict_instance = strict_instance;
    c.strict_witness = strict_witness;
    c.strict_carry = strict_carry;
    c.raw_source = raw_source;
    c.raw_const_of = raw_const_of;
    c.strand_level_ptr = strand_level_ptr;
    c.window_first_op = window_first_op;
    c.n_slots = n_slots;
    c.n_levels = n_levels;
    c.max_level_width = max_level_width;
    c.retain_all = retain_all;
    c.boolean_path = boolean_path;
    c.has_bitops = has_bitops;
    c.const_words = const_words;
    c.words_per_const = words_per_const;
    return c;
  }
};

// Ops [lo, hi) of a tape.  Arrays are indexed by (tape index - lo); operands are tape indices and may lie below lo.
struct TapeWindow {
  uint32_t lo = 0, hi = 0;
  const uint8_t* kind = nullptr;
  const uint32_t* a = nullptr;
  const uint32_t* b = nullptr;
  const uint32_t* drops = nullptr;        // handles dropped since the previous window, up to tape position hi
  size_t n_drops = 0;
  const Tape::Ladder* ladders = nullptr;  // ladder hints that end inside the window
  size_t n_ladders = 0;
  bool final = false;                     // last window: every value not in `pinned` has seen its last reader
  const std::vector<uint32_t>* pinned = nullptr;
  const std::vector<Value>* consts = nullptr;   // the tape's constant pool (GF(2): the parity of a constant decides what add_constant / mul_constant become)
  const std::vector<uint8_t>* const_parity = nullptr;   // ... or just the low bit of every constant of the pool so far (a streamed
                                                        // window: the pool itself keeps growing on the recording thread)
};

struct WindowResult {  // what add_window() appended to the schedule
  uint64_t first_op = 0, n_ops = 0;
  uint32_t first_launch = 0, n_launches = 0;
};

class StreamScheduler {
 public:
  StreamScheduler(const FieldHost& field, const ScheduleOptions& opt);
  ~StreamScheduler();
  StreamScheduler(const StreamScheduler&) = delete;
  StreamScheduler& operator=(const StreamScheduler&) = delete;
  WindowResult add_window(const TapeWindow& w);
  // after the final window: the constant pool in device form; returns the finished schedule
  Schedule finish(const std::vector<Value>& consts);
  const Schedule& partial() const;   // the program so far (ops / launches of the windows already added)

 private:
  struct Impl;
  Impl* impl_;
};

// the whole tape as one final window
Schedule build_schedule(const Tape& tape, const FieldHost& field, const ScheduleOptions& opt);
// the tape cut at tape.cuts (what a streamed ingest produces window by window), for tests and tools
Schedule build_schedule_windowed(const Tape& tape, const FieldHost& field, const ScheduleOptions& opt);

}  // namespace zki
