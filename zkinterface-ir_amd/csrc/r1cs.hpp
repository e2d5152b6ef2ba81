// R1CS side of the path (BASELINE config #5).
//
// The reference's `ToR1CSConverter` (rust/src/consumers/to_r1cs.rs:12-393) is a ZKBackend that
// answers every backend call with a zkinterface variable and one BilinearConstraint A*B=C, and (with
// use_witness) computes the assignment of every variable, i.e. the plaintext evaluation.  Here the
// calls are already on the tape, so the constraint system is derived from the tape with the same
// per-op rules, the assignment is what the replay kernels leave in the wire table, and a HIP kernel
// checks <a,w>*<b,w> = <c,w> for every row and every witness lane.
#pragma once
#include <stdint.h>
#include <vector>

#include "tape.hpp"

namespace zki {

constexpr uint64_t kVarOne = 0;  // "self.one = 0; // spec convention" (to_r1cs.rs:117)

struct R1csTerm {
  uint64_t var;    // converter variable id (or, for loaded CSR, the caller's id space)
  uint32_t coef;   // index into R1cs::coefs
};

struct R1cs {
  uint64_t n_vars = 1;                 // ids 0 .. n_vars-1; 0 is the constant one
  std::vector<Value> coefs;            // distinct little-endian coefficient strings, as the reference writes them
  std::vector<uint32_t> row_ptr;       // 3 entries per row (start of A, B, C) + one final end marker
  std::vector<R1csTerm> terms;
  std::vector<uint64_t> var_of_op;     // per tape op: its variable (kNoVar for assert_zero)
  std::vector<uint8_t> var_kind;       // per variable: 0 one, 1 instance (incl. constants), 2 witness, 3 internal, 4 correction
  size_t n_rows() const { return row_ptr.empty() ? 0 : (row_ptr.size() - 1) / 3; }
};
constexpr uint64_t kNoVar = ~0ull;

// to_r1cs.rs:143-393 applied to the recorded calls, in order.
R1cs r1cs_from_tape(const Tape& tape, const FieldHost& field, const Value& modulus, bool use_correction);

// device form of one row (device/r1cs_kernels.hpp)
struct R1csRowDev {
  uint32_t first;   // first term
  uint32_t counts;  // nA | nB << 8 | nC << 16 | flags << 24: 1 = B is the constant one; bits 1-2 / 3-4 / 5-6: the
                    // coefficient class of A / B / C
};
struct R1csTermDev {
  uint32_t slot;    // wire-table slot; 0xFFFFFFFF = the constant one
  uint32_t coef;    // index into the Montgomery coefficient pool; 0xFFFFFFFF = coefficient 1.  In a combination of class
                    // unit / small: the coefficient as a signed integer, sign << 31 | magnitude
};
// coefficient classes of a combination (== zkgpu::kR1csClass* of device/args.hpp, which says what each one costs)
constexpr uint32_t kR1csClassFull = 0, kR1csClassUnit = 1, kR1csClassSmall = 2;
constexpr uint32_t kR1csClassShiftA = 1, kR1csClassShiftB = 3, kR1csClassShiftC = 5;

}  // namespace zki
