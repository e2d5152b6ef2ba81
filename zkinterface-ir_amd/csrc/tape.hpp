// TapeBackend: a ZKBackend (rust/src/consumers/evaluator.rs:17-76) whose Wire
// is a handle into a linear gate tape.  Semantic twin of the reference's
// IRFlattener (rust/src/consumers/flattening.rs:42-191), which also answers
// every backend call with a freshly numbered wire; here the calls are kept in
// memory as the program the HIP kernels replay for a whole batch of witnesses.
#pragma once
#include <stdint.h>
#include <map>
#include <string>
#include <vector>

#include "sieve/structs.hpp"

namespace zki {

// op kinds shared with the device (device/replay_kernels.hpp OpKind)
enum TapeKind : uint8_t {
  TK_NOP = 0, TK_ADD = 1, TK_MUL = 2, TK_ADDC = 3, TK_MULC = 4, TK_COPY = 5, TK_CONST = 6,
  TK_INSTANCE = 7, TK_WITNESS = 8, TK_ASSERT = 9, TK_AND = 10, TK_XOR = 11, TK_NOT = 12,
  TK_NZ = 13,  // scheduler only: 1 if the operand is non-zero else 0 (= x^(p-1) for a prime p), never recorded
  TK_CARRY = 14,  // a value carried over from the previous field segment of the session (a: its index in the carry
                  // stream).  Not a backend call of the reference: the wire simply lived on when the modulus changed
                  // (evaluator.rs:232-237); a source like Instance / Witness, holding the UNREDUCED integer
  // scheduler only (strands, schedule.cpp): an Instance / Witness entry taken apart -- the value as it lies in the input
  // buffer copied into an LDS value of the strand (a: position, a1: 0 instance / 1 witness), and its conversion from there
  // (a: that LDS value, a1: the stream, b: the position) one or more levels later
  TK_INPUT_RAW = 15, TK_INPUT_CONV = 16,
};
const char* tape_kind_name(uint8_t k);  // names of SURVEY.md Appendix A ("copy", "mul", ...)

constexpr uint32_t kNoWire = 0xFFFFFFFFu;

constexpr int kFieldWords = 128;  // 32-bit words of the widest supported field characteristic (4096 bits)
constexpr int kMontWords = 16;    // ... of the widest one the Montgomery kernels take (512 bits, device/fp_mont.hpp)

// Host-side description of the field: limbs and the constants the device needs, plus canonicalisation of
// arbitrary-length little-endian Values.  Three representations of a wire on the device:
//   * is_two   -- p == 2, one bit per witness (device/bool_kernels.hpp);
//   * default  -- odd p < 2^512: Montgomery form (device/fp_mont.hpp FieldParams: r2, one, n0inv);
//   * generic  -- every other modulus >= 2 up to 4096 bits (even, wider than 512 bits), and any modulus when asked for
//                 (GF(2) in a session that also works in another field): canonical residues, Barrett reduction with
//                 `mu` (device/generic_kernels.hpp GenericParams).
struct FieldHost {
  uint32_t nwords = 0;          // 32-bit words per wire value: 2, 4, ... (64-bit limb granularity)
  uint32_t bits = 0;
  bool is_two = false;          // p == 2 on the bit-packed path
  bool generic = false;         // canonical residues + Barrett (the any-modulus kernels)
  uint32_t kwords = 0;          // generic: 32-bit words of p, the top one non-zero
  uint32_t p[kFieldWords] = {0}, r2[kFieldWords] = {0}, one[kFieldWords] = {0};
  uint32_t mu[kFieldWords + 2] = {0};   // generic: floor(2^(64 kwords) / p)
  uint32_t n0inv = 0;

  void init(const Value& modulus_le, bool force_generic = false);   // throws zki::Error if unsupported
  bool p_is_two() const { return bits == 2 && p[0] == 2; }    // whatever the representation
  bool is_canonical(const Value& v) const;                    // v < p as integers
  void reduce(const Value& v, uint32_t out[kFieldWords]) const;         // v mod p
  // the form constants and inputs take in the wire table: in * R mod p (Montgomery), or `in` itself (generic)
  void to_mont(const uint32_t in[kFieldWords], uint32_t out[kFieldWords]) const;
  void add(const uint32_t a[kFieldWords], const uint32_t b[kFieldWords], uint32_t out[kFieldWords]) const;
};

struct Tape {
  // one entry per backend call, in call order; asserts included (no value)
  std::vector<uint8_t> kind;
  std::vector<uint32_t> a, b;      // operand handles; const-pool index; input position; assert seq
  // asserts, in execution order
  std::vector<uint32_t> assert_op;     // tape index of the k-th assert_zero
  std::vector<uint64_t> assert_wire;   // local id printed in "Wire_{} (may be weighted) ..."
  // constant pool: distinct little-endian byte strings
  std::vector<Value> consts;
  // Exponent ladders of Switch weights (evaluator.rs:801-839): ops [first, result] compute
  // result = base^(modulus - 1) and nothing outside reads any of them but `result`.
  struct Ladder {
    uint32_t first, result, base;
  };
  std::vector<Ladder> ladders;
  uint32_t ladder_open = kNoWire;          // tape size at note_ladder_begin while a ladder is being recorded
  uint32_t n_instance = 0, n_witness = 0;  // input positions referenced (max + 1)
  uint32_t n_carry = 0;                    // values carried in from the previous field segment
  uint64_t n_value_ops = 0;
  uint32_t n_rebound = 0;                  // the first n_rebound entries re-bind the wires an earlier field segment left alive
                                           // (carried values, inputs and constants read again): no backend calls
  // Dropped wires, in order: handle drop_handle[k] went out of the caller's reach when the tape held drop_pos[k]
  // entries -- no call recorded at or after that position can name it.  The reference's evaluator owns its wires
  // (`HashMap<WireId, B::Wire>`, temporaries in locals) and Rust drops them on `Free`, at scope exit and at the end
  // of each expression (evaluator.rs:698-746, :775-797); the streaming scheduler needs exactly that signal to know
  // that a value has seen its last reader before the rest of the relation has arrived.
  std::vector<uint32_t> drop_pos, drop_handle;
  // Window cuts for the streaming scheduler, decided while recording so that they depend on the tape alone (not on
  // how the relation was split into messages): the first position >= the previous cut + window_ops at which no
  // exponent ladder is open.
  std::vector<uint32_t> cuts;
  uint32_t window_ops = 0;                 // 0 = no cuts

  size_t size() const { return kind.size(); }
};

// FieldElement of the recording backend: either literal bytes (constants) or a
// position in the per-lane instance / witness stream.
struct TapeElement {
  enum Kind : uint8_t { LITERAL, INSTANCE_REF, WITNESS_REF } kind = LITERAL;
  uint32_t position = 0;
  Value bytes;
};

class TapeBackend;

// The recording backend's `Wire`: owns one tape handle, move-only like the reference's `B::Wire` values.  Letting go of
// it (scope erase on `Free`, a sub-circuit scope going away, a temporary dying) tells the backend that no later call
// can read the value.
struct TapeWire {
  uint32_t h = kNoWire;
  TapeBackend* owner = nullptr;
  TapeWire() = default;
  TapeWire(uint32_t handle, TapeBackend* o) : h(handle), owner(o) {}
  TapeWire(TapeWire&& o) noexcept : h(o.h), owner(o.owner) { o.h = kNoWire; o.owner = nullptr; }
  TapeWire& operator=(TapeWire&& o) noexcept {
    if (this != &o) {
      release();
      h = o.h;
      owner = o.owner;
      o.h = kNoWire;
      o.owner = nullptr;
    }
    return *this;
  }
  TapeWire(const TapeWire&) = delete;
  TapeWire& operator=(const TapeWire&) = delete;
  ~TapeWire() { release(); }
  inline void release();
};

class TapeBackend {
 public:
  using Wire = TapeWire;
  using FieldElement = TapeElement;

  static FieldElement from_bytes_le(const Value& v) {  // evaluator.rs:25
    TapeElement e;
    e.bytes = v;
    return e;
  }
  void set_field(const Value& modulus, uint32_t degree, bool is_boolean);
  static FieldElement literal_bytes(const Value& bytes) {   // a constant as its little-endian bytes
    TapeElement e;
    e.bytes = bytes;
    return e;
  }
  FieldElement one() const { return literal(1); }
  FieldElement minus_one() const;
  FieldElement zero() const { return literal(0); }

  // ---- the trait, on owned wires (what Evaluator<TapeBackend> calls) ----
  Wire copy(const Wire& w) { return own(h_copy(w.h)); }
  Wire constant(FieldElement val) { return own(h_constant(std::move(val))); }
  void assert_zero(const Wire& w) { h_assert_zero(w.h); }
  Wire add(const Wire& x, const Wire& y) { return own(arith(TK_ADD, x.h, y.h)); }
  Wire multiply(const Wire& x, const Wire& y) { return own(arith(TK_MUL, x.h, y.h)); }
  Wire add_constant(const Wire& x, FieldElement c) { return own(h_add_constant(x.h, std::move(c))); }
  Wire mul_constant(const Wire& x, FieldElement c) { return own(h_mul_constant(x.h, std::move(c))); }
  Wire and_(const Wire& x, const Wire& y) { return own(bitwise(TK_AND, x.h, y.h)); }
  Wire xor_(const Wire& x, const Wire& y) { return own(bitwise(TK_XOR, x.h, y.h)); }
  Wire not_(const Wire& x) { return own(bitwise(TK_NOT, x.h, 0)); }
  Wire instance(FieldElement val) { return own(h_instance(std::move(val))); }
  Wire witness(const FieldElement* val) { return own(h_witness(val)); }
  void note_assert_wire(WireId local_id) { pending_assert_wire_ = local_id; }
  size_t note_ladder_begin() {
    tape_.ladder_open = (uint32_t)tape_.size();
    return tape_.size();
  }
  void note_ladder_end(size_t first, const Wire& base, const Wire& result) { h_ladder(first, base.h, result.h); }

  // ---- the same on plain handles (the C ABI: the caller owns the handles and reports drops itself) ----
  uint32_t h_copy(uint32_t w) { return push(TK_COPY, w, 0); }
  uint32_t h_constant(FieldElement val);
  void h_assert_zero(uint32_t w);
  uint32_t h_add(uint32_t x, uint32_t y) { return arith(TK_ADD, x, y); }
  uint32_t h_multiply(uint32_t x, uint32_t y) { return arith(TK_MUL, x, y); }
  uint32_t h_add_constant(uint32_t x, FieldElement c);
  uint32_t h_mul_constant(uint32_t x, FieldElement c);
  uint32_t h_and(uint32_t x, uint32_t y) { return bitwise(TK_AND, x, y); }
  uint32_t h_xor(uint32_t x, uint32_t y) { return bitwise(TK_XOR, x, y); }
  uint32_t h_not(uint32_t x) { return bitwise(TK_NOT, x, 0); }
  uint32_t h_instance(FieldElement val);
  uint32_t h_witness(const FieldElement* val);
  uint32_t h_carry(uint32_t index);
  // a wire that holds instance / witness value `position` again, without advancing a stream: the integer a wire of the
  // previous field segment still held when the field changed (capi.cpp switch_field)
  uint32_t h_input_at(uint8_t kind, uint32_t position);   // a wire of the previous field segment, alive in the scope when the modulus changed
  void h_ladder(size_t first, uint32_t base, uint32_t result) {
    tape_.ladder_open = kNoWire;
    // with is_boolean the "multiplies" of the ladder are `and` gates and Fermat says nothing about them
    // (evaluator.rs:86-93): no hint is kept
    if (!is_boolean_ && result >= first && base < first) tape_.ladders.push_back({(uint32_t)first, result, base});
    maybe_cut();   // a cut that fell inside the ladder was put off until here
  }
  // no call recorded from now on reads `h` (TapeWire's destructor; zkgpu_backend_drop)
  void drop_wire(uint32_t h) {
    if (h >= tape_.size()) return;
    tape_.drop_pos.push_back((uint32_t)tape_.size());
    tape_.drop_handle.push_back(h);
  }
  // streaming: cut the tape into windows of about `ops` entries; `hook(arg)` runs on the recording thread at each cut
  void set_window(uint32_t ops, void (*hook)(void*), void* arg) {
    tape_.window_ops = ops;
    cut_hook_ = hook;
    cut_arg_ = arg;
    if (!ops) {
      depth_.clear();
      depth_.shrink_to_fit();
    } else if (depth_.size() < tape_.size()) {
      depth_.resize(tape_.size(), 0);   // (set before the first Relation message: nothing recorded yet)
    }
  }

  // Single-statement use (`evaluate <workspace>`): the values of an Instance /
  // Witness message become lane 0's input stream, referenced by position.
  FieldElement import_instance(const Value& v);
  FieldElement import_witness(const Value& v);
  static FieldElement instance_ref(uint32_t position);
  static FieldElement witness_ref(uint32_t position);

  const Tape& tape() const { return tape_; }
  // Loops are unrolled into the tape; a limit keeps a corrupt or hostile bound from exhausting the host.
  void set_max_ops(uint64_t n) { max_ops_ = n; }
  const FieldHost& field() const { return field_; }
  const Value& modulus() const { return modulus_; }  // bytes given to set_field
  bool field_set() const { return field_set_; }
  bool is_boolean() const { return is_boolean_; }
  const std::vector<Value>& lane0_instances() const { return lane0_instances_; }
  const std::vector<Value>& lane0_witnesses() const { return lane0_witnesses_; }
  // a new field segment continues the input streams of the one before it (positions are global)
  void adopt_streams(TapeBackend& from) {
    lane0_instances_ = std::move(from.lane0_instances_);
    lane0_witnesses_ = std::move(from.lane0_witnesses_);
    from.lane0_instances_.clear();
    from.lane0_witnesses_.clear();
    max_ops_ = from.max_ops_;
  }
  // A session that works in GF(2) AND in another field keeps every wire as an integer (the any-modulus kernels), GF(2)
  // included: bit-packed wires cannot be carried over (capi.cpp switch_field).  The recorded tape does not depend on the
  // representation, so a backend that has recorded GF(2) gates already can still be told.
  void use_generic_field() {
    force_generic_ = true;
    if (field_set_) field_.init(modulus_, true);
  }
  // everything recorded so far re-binds wires of the previous field segment (capi.cpp switch_field)
  void end_rebinding() {
    tape_.n_rebound = (uint32_t)tape_.size();
    tape_.n_value_ops = 0;
  }
  void set_assert_base(uint32_t n) { assert_base_ = n; }   // global sequence number of this segment's first assert
  uint32_t assert_base() const { return assert_base_; }

 private:
  static FieldElement literal(uint8_t v) {
    TapeElement e;
    e.bytes.assign(1, v);
    return e;
  }
  Wire own(uint32_t h) { return Wire(h, this); }
  uint32_t push(uint8_t kind, uint32_t a, uint32_t b);
  uint32_t arith(uint8_t kind, uint32_t a, uint32_t b);
  uint32_t bitwise(uint8_t kind, uint32_t a, uint32_t b);
  // the fallback cut: a window twice as long as asked for that has not met a level seam (note_level, tape.cpp) ends here
  void maybe_cut() {
    if (!tape_.window_ops || tape_.ladder_open != kNoWire) return;
    const uint32_t last = tape_.cuts.empty() ? 0 : tape_.cuts.back();
    if (tape_.size() - last < 2 * (uint64_t)tape_.window_ops) return;
    tape_.cuts.push_back((uint32_t)tape_.size());
    if (cut_hook_) cut_hook_(cut_arg_);
  }
  void note_level(uint8_t kind, uint32_t a, uint32_t b);
  std::vector<uint32_t> depth_;   // streaming: dependency depth per entry
  uint32_t top_depth_ = 0;
  void (*cut_hook_)(void*) = nullptr;
  void* cut_arg_ = nullptr;
  uint32_t intern(const Value& bytes);
  void need_field() const;

  Tape tape_;
  FieldHost field_;
  bool field_set_ = false, is_boolean_ = false, force_generic_ = false;
  Value modulus_;
  std::map<Value, uint32_t> const_index_;
  WireId pending_assert_wire_ = 0;
  uint32_t assert_base_ = 0;
  uint64_t max_ops_ = 1ull << 30;
  std::vector<Value> lane0_instances_, lane0_witnesses_;
};

inline void TapeWire::release() {
  if (owner && h != kNoWire) owner->drop_wire(h);
  owner = nullptr;
  h = kNoWire;
}

}  // namespace zki
