// Records shared by the host engine and the gfx950 kernels: program entries, kernel argument blocks and the
// host-callable launchers.  The kernels themselves live in the other headers of this directory and are compiled
// in translation units of their own (kernels_arith.hip once per field width, kernels_bool.hip), so that the
// engine's host code never instantiates a kernel and the widths build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lds_layout.hpp"

namespace zkgpu {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int kMaxWords = 16;  // fields up to 512 bits

// Per-field constants, passed by value in the kernarg segment (wave-uniform:
// the compiler keeps them in SGPRs).
struct FieldParams {
  u32 p[kMaxWords];    // modulus, little-endian 32-bit words
  u32 r2[kMaxWords];   // R^2 mod p   (to_mont multiplier)
  u32 one[kMaxWords];  // R mod p     (Montgomery form of 1)
  u32 n0inv;           // -p^{-1} mod 2^32
  u32 nwords;          // N actually used (2, 4, ..., 16)
  // conditional subtractions that make the lazily reduced sum of K Montgomery products canonical: the sum is below
  // (K * p / R + 1) * p, so ceil(K * p / R) of them (host: Engine::load_program); index K - 1, K = 1..4
  u32 dot_rounds[4];
  // != 0: the sum of 3 products may stay unreduced when it is an operand of a Montgomery product of two such sums:
  // with a = 3 * p / R + 1 the sum is below a * p < R and the product below (a * a * p / R + 1) * p <= 2 * p, which
  // the product's own conditional subtraction makes canonical (BN254: a = 1.57, a * a * p / R = 0.46)
  u32 lazy_dot3;
};

// Any-modulus path (generic_kernels.hpp): canonical residues, Barrett reduction.  Lives in device memory (520 words are
// too many for the kernarg block); filled by Engine::load_program from FieldHost.
constexpr int kGenericMaxWords = 128;   // characteristics up to 4096 bits
struct GenericParams {
  u32 k;                            // 32-bit words of p, the top one non-zero
  u32 nwords;                       // words per wire value: 2 * ceil(bits / 64), k or k + 1
  u32 p[kGenericMaxWords];
  u32 mu[kGenericMaxWords + 2];     // floor(2^(64 k) / p): k + 2 words (the last is 0 unless p is a power of 2^32)
  u32 pow2_bits;                    // B where p = 2^B (arithmetic of a ring Z / 2^B: x mod p is the low B bits, no Barrett), else 0
};

enum OpKind : u32 {
  OP_NOP = 0,
  OP_ADD = 1,       // dst = a + b
  OP_MUL = 2,       // dst = a * b
  OP_ADDC = 3,      // dst = a + const[b]
  OP_MULC = 4,      // dst = a * const[b]
  OP_COPY = 5,      // dst = a
  OP_CONST = 6,     // dst = const[a]
  OP_INSTANCE = 7,  // dst = to_mont(instance[lane][a])
  OP_WITNESS = 8,   // dst = to_mont(witness[lane][a])
  OP_ASSERT = 9,    // a must be zero; b = assert sequence number
  OP_AND = 10,      // and / xor / not: bits for p = 2 (bool_kernels.hpp); integer bit ops then % p for an odd p
  OP_XOR = 11,
  OP_NOT = 12,
  OP_NZ = 13,  // 1 if the operand is non-zero else 0: x^(p-1) over a prime field (scheduler-made, schedule.cpp)
  OP_CARRY = 14,  // dst = to_mont(carry[lane][a]): a wire of the previous field segment, as the integer it held
  // strands only (schedule.cpp): an instance / witness entry in two halves, so that neither is the longest entry of its level
  OP_INPUT_RAW = 15,   // LDS value dst = the words of stream a1 (0 instance, 1 witness), position a0, as they lie in the buffer
  OP_INPUT_CONV = 16,  // dst = what OP_INSTANCE / OP_WITNESS makes of position b0 of stream a1, the words taken from LDS value a0
};

struct TapeOp {
  u32 dst;
  u32 a;
  u32 b;
  u32 kind;
};

// 32-byte program entry of the fused schedule (host: DevOp2, schedule.hpp)
struct TapeOp2 {
  u32 dst, kind, a0, a1, b0, b1, pad0, pad1;
};

constexpr u32 kNoFail = 0xFFFFFFFFu;
// an operand of `and` / `xor` over a field other than GF(2) that names the RAW value of an input instead of a wire-table
// slot: kOperandIsSource | (2 + 4 * position + stream), the code of the assert_zero / not sinks (schedule.cpp)
constexpr u32 kOperandIsSource = 0x80000000u;
constexpr u32 kLaneFlagNonCanonical = 1u;
// a slot number of a STRAND's entry with this bit names value k of the workgroup's LDS, [k][chunk][lane] (host: schedule.hpp)
constexpr u32 kSlotInLds = 0x40000000u;

// What only the input arms of the replay kernels read (instance / witness / carried values and their modes): kept behind
// ONE pointer so that the eleven words do not sit in SGPRs through every other arm of the cold kernels (with them in the
// kernarg block the cold kernels spilled 28 SGPRs and the strands of a structured relation ran 13 % slower).
struct InputAux {
  const uint8_t* strict_inst;   // per input position: 0xFF = a value >= p flags the lane (Schedule::strict_instance)
  const uint8_t* strict_wit;
  const uint8_t* strict_carry;
  const u32* carry;             // [lane][n_carry][carry_words]: canonical values of the wires carried over from the previous
  u32 n_carry, carry_words;     // field segment (carry_words = that field's N)
  u32 in_stride_words;          // 32-bit words per input value in inst / wit (N, or more in a session of several fields)
  u32 raw_const_base;           // stream 3 of the source codes: constants >= p whose bits are read, kept as plain integers in
                                // the constant pool from this entry on (Schedule::raw_const_of)
  // developer instrumentation (null in production): the strand kernel of lane block 0 writes clock stamps here,
  // [level < kStampLevels][wave 0..3][4]: level start, entry fetched, entry done, barrier passed (tools/strand_stamps.py)
  unsigned long long* stamps;
};

struct ReplayArgs {
  const TapeOp* ops;      // ops of this launch (device)
  u32 n_ops;
  u32 ops_per_wave;       // contiguous ops walked by one wave
  uint4* table;           // wire table
  u32 n_slots;            // slots per lane block
  u32 batch;              // real lanes
  u32 lb_base;            // first lane block of this launch (lane groups)
  const u32* consts;      // constant pool, Montgomery form, N words each
  const uint8_t* inst;    // [lane][n_inst][4N bytes] little-endian, canonical
  const uint8_t* wit;     // [lane][n_wit][4N bytes]
  u32 n_inst;
  u32 n_wit;
  u32* first_fail;        // [lane] min assert sequence number that failed
  u32* lane_flags;        // [lane] sticky flags (non-canonical input ...)
  u32 xcd_chunks;         // != 0: XCD-aware 1-D grid, see block_coords()
  const InputAux* aux;    // device memory, wave-uniform: read on the scalar path
};

struct ReplayArgs2 {
  const TapeOp2* ops;
  u32 n_ops;
  u32 ops_per_wave;
  uint4* table;
  u32 n_slots;
  u32 batch;
  u32 lb_base;
  const u32* consts;
  const uint8_t* inst;
  const uint8_t* wit;
  u32 n_inst;
  u32 n_wit;
  u32* first_fail;
  u32* lane_flags;
  u32 xcd_chunks;
  u32 op_stride;          // 1 or 4, see the kernel; a strand launch: 1 = the input buffers hold values of exactly N words
  const InputAux* aux;    // as in ReplayArgs
};
constexpr u32 kStampLevels = 256;

// instantiations of replay_fused_kernel (replay_kernels.hpp)
constexpr int kFusedHot = 0, kFusedMisc = 1, kFusedAll = 2;

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

// (LdsOp, the chunk and block headers and the row constants of the LDS-resident kernel: lds_layout.hpp)
struct BoolLdsArgs {
  const LdsOp* ops;         // generic chunks (inputs, constants, asserts, sequential segments): 8-byte entries
  const u32* ops6;          // rows: 12 bytes per thread and row (two ops of three u16 each)
  const u32* blocks;        // block headers
  u32 block_rows;           // rows per block of this program = rows every block fetches
  const u32* chunks;
  u32 n_chunks;
  u32 n_slots;              // including the kLdsExtraSlots scratch / constant slots
  u32 batch;
  u32 n_cols;               // 32-witness slices in the batch
  u32 total_words64;        // 64 * lane blocks (layout of the packed inputs)
  const u32* consts;
  const u32* packed_inst;   // u32 view of packed[position][word64]
  const u32* packed_wit;
  u32* first_fail;
  u64* table;               // HBM table (bool_replay_kernel layout) for the optional write-back
  u32 writeback;
};

struct R1csRow {
  u32 first;
  u32 counts;  // nA | nB << 8 | nC << 16 | flags << 24
};
struct R1csTerm {
  u32 slot;  // 0xFFFFFFFF: the constant one
  u32 coef;  // index into the coefficient pool, 0xFFFFFFFF: coefficient 1.  In a combination of class unit / small:
             // sign << 31 | magnitude of the coefficient as a signed integer (magnitude < 2^31)
};
constexpr u32 kR1csBIsOne = 1u;
// Coefficient classes of a row's three combinations (row flags bits 1-2: A, 3-4: B, 5-6: C), set by the host
// (capi.cpp build_device_rows) -- what FromR1CSConverter expansions and hand-written R1CS mostly hold
// (from_r1cs.rs:110-125) are coefficients like 1, -1, 2, 2^k:
//   full  -- any coefficients: Montgomery products with pool entries (the terms whose coefficient is 1 are added)
//   unit  -- every coefficient is +1 or -1 (at least one -1): additions and subtractions only
//   small -- every coefficient is a signed integer below 2^31 in magnitude: N word products per term instead of N^2
//            and two Montgomery word rounds per combination (r1cs_lincomb_small)
constexpr u32 kR1csClassFull = 0u, kR1csClassUnit = 1u, kR1csClassSmall = 2u;
constexpr u32 kR1csClassShiftA = 1u, kR1csClassShiftB = 3u, kR1csClassShiftC = 5u;

struct R1csArgs {
  const R1csRow* rows;
  const R1csTerm* terms;
  const u32* coefs;      // Montgomery form, N words each
  u32 first_row, n_rows; // rows [first_row, first_row + n_rows) of this launch
  const uint4* table;    // read side
  uint4* table_out;      // ASSIGN: same table
  u32 n_slots;
  u32 batch;
  u32* first_fail;       // CHECK: min failing row per lane
  u32 one_coef;          // index of the Montgomery form of 1 in `coefs` (the host appends it to the pool); behind it
                         // the Montgomery forms of 2^64 and 2^128 (what undoes the word rounds of small-class sums)
};

// quotient wires of the R1CS conversion (r1cs_correction_kernel)
struct R1csCorrCall {
  u32 a, b, out;   // slots of the operands and of the result; b = constant index with kCorrConstB
  u32 flags;
};
constexpr u32 kCorrMul = 1u, kCorrConstB = 2u;
struct R1csCorrArgs {
  const R1csCorrCall* calls;
  u32 n_calls;
  const uint4* table;
  u32 n_slots;
  u32 batch;
  const u32* consts;   // canonical little-endian words of the raw constants, N each
  u32 pinv[kMaxWords]; // p^{-1} mod 2^(32N)
  u32* out;            // [lane][call][N words]
};

// ---- launchers (defined in kernels_arith.hip, one set per field width, and kernels_bool.hip) ----
#define ZKGPU_DECLARE_WIDTH(W)                                                                                      \
  void launch_replay_fused_w##W(int cls, dim3 grid, size_t lds_pad, hipStream_t st, const ReplayArgs2& a,         \
                                const FieldParams& fp);                                                             \
  void launch_replay_strand_w##W(int cls, dim3 grid, hipStream_t st, const ReplayArgs2& a, const u32* level_ptr,   \
                                 u32 n_levels, size_t lds_bytes, const FieldParams& fp);                           \
  void launch_replay_w##W(bool bitops, dim3 grid, hipStream_t st, const ReplayArgs& a, const FieldParams& fp);     \
  void launch_r1cs_w##W(bool assign, bool classes, dim3 grid, hipStream_t st, const R1csArgs& a, const FieldParams& fp);         \
  void launch_dump_w##W(dim3 grid, hipStream_t st, const uint4* table, u32 n_slots, const u32* slots, u32 n_dump,  \
                        u32 batch, u32* out, const FieldParams& fp);                                               \
  void launch_r1cs_corr_w##W(dim3 grid, hipStream_t st, const R1csCorrArgs& a, const FieldParams& fp);
ZKGPU_DECLARE_WIDTH(2)
ZKGPU_DECLARE_WIDTH(4)
ZKGPU_DECLARE_WIDTH(6)
ZKGPU_DECLARE_WIDTH(8)
ZKGPU_DECLARE_WIDTH(10)
ZKGPU_DECLARE_WIDTH(12)
ZKGPU_DECLARE_WIDTH(14)
ZKGPU_DECLARE_WIDTH(16)
#undef ZKGPU_DECLARE_WIDTH

// any-modulus path (kernels_generic.hip)
void launch_replay_generic(dim3 grid, hipStream_t st, const ReplayArgs& a, const GenericParams* gp_device, u32 nwords, u32 k_words);
void launch_dump_generic(dim3 grid, hipStream_t st, const uint4* table, u32 n_slots, const u32* slots, u32 n_dump, u32 batch,
                         u32* out, u32 nwords);
// the arithmetic of that path run on the HOST (the same functions): op 0 add, 1 mul, 2 reduce(a), 3 and, 4 xor; a, b, out
// hold gp->nwords words (op 2: a holds gp->nwords raw words).  Returns 0, or 1 for an unknown op.
int generic_selftest(const GenericParams* gp, int op, const u32* a, const u32* b, u32* out);

void launch_verdict(dim3 grid, hipStream_t st, const u32* first_fail, const u32* lane_flags, u32 batch,
                    unsigned long long* counts);
void launch_pack_inputs(hipStream_t st, const uint8_t* inst, u32 n_inst, const uint8_t* strict_inst, u64* packed_inst,
                        const uint8_t* wit, u32 n_wit, const uint8_t* strict_wit, u64* packed_wit, u32 batch, u32 total_words,
                        u32* lane_flags);
void launch_bool_replay(dim3 grid, hipStream_t st, const BoolReplayArgs& a);
hipError_t bool_lds_set_max_shared(int bytes);
bool bool_lds_has_block_rows(u32 rows);
void launch_bool_lds(u32 n_cols, size_t lds_bytes, hipStream_t st, const BoolLdsArgs& a);
void launch_bool_dump(dim3 grid, hipStream_t st, const u64* table, u32 n_slots, const u32* slots, u32 n_dump,
                      u32 batch, uint8_t* out);

}  // namespace zkgpu
