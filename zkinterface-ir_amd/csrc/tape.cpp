#include "tape.hpp"

#include <algorithm>

#include <string.h>

namespace zki {

const char* tape_kind_name(uint8_t k) {
  static const char* names[] = {"nop", "add", "mul", "addc", "mulc", "copy", "constant",
                                "instance", "witness", "assert_zero", "and", "xor", "not"};
  return k <= TK_NOT ? names[k] : k == TK_NZ ? "nz" : k == TK_CARRY ? "carry" : "?";
}

// ---------------------------------------------------------------- FieldHost
namespace {
constexpr int W = kFieldWords;
bool geq_words(const uint32_t* a, const uint32_t* b, int n) {
  for (int i = n - 1; i >= 0; --i)
    if (a[i] != b[i]) return a[i] > b[i];
  return true;
}
void sub_words(uint32_t* a, const uint32_t* b, int n) {
  uint64_t borrow = 0;
  for (int i = 0; i < n; ++i) {
    const uint64_t d = (uint64_t)a[i] - b[i] - borrow;
    a[i] = (uint32_t)d;
    borrow = (d >> 63) & 1;
  }
}
size_t significant_bytes(const Value& v) {
  size_t n = v.size();
  while (n > 0 && v[n - 1] == 0) --n;
  return n;
}
}  // namespace

void FieldHost::add(const uint32_t a[W], const uint32_t b[W], uint32_t out[W]) const {
  // (operands are below p, so only the words in use carry anything; the rest of `out` is cleared)
  const int n = (int)nwords;
  uint32_t r[W];
  uint64_t c = 0;
  for (int i = 0; i < n; ++i) {
    c += (uint64_t)a[i] + b[i];
    r[i] = (uint32_t)c;
    c >>= 32;
  }
  if (c || geq_words(r, p, n)) sub_words(r, p, n);
  memcpy(out, r, 4 * (size_t)n);
  if (n < W) memset(out + n, 0, 4 * (size_t)(W - n));
}

void FieldHost::init(const Value& modulus_le, bool force_generic) {
  *this = FieldHost();
  const size_t n = significant_bytes(modulus_le);
  if (n == 0) throw Error("Modulus cannot be zero.");  // evaluator.rs:868-869
  if (n > 4 * (size_t)W) throw Error("GPU backend: field characteristic wider than 4096 bits is not supported");
  for (size_t i = 0; i < n; ++i) p[i / 4] |= (uint32_t)modulus_le[i] << (8 * (i % 4));
  for (int i = W - 1; i >= 0 && bits == 0; --i)
    if (p[i]) bits = 32 * i + (32 - __builtin_clz(p[i]));
  if (bits < 2) throw Error("GPU backend: field characteristic 1 is not supported");
  nwords = 2 * ((bits + 63) / 64);
  if (p_is_two() && !force_generic) {
    is_two = true;
    return;
  }
  if (force_generic || (p[0] & 1) == 0 || bits > 32 * kMontWords) {
    // canonical residues, Barrett reduction: mu = floor(b^(2k) / p), b = 2^32, by long division bit by bit
    generic = true;
    kwords = (bits + 31) / 32;
    one[0] = 1;
    uint32_t rem[W + 1] = {0};
    const int rw = (int)kwords + 1;
    uint32_t pw[W + 1] = {0};
    memcpy(pw, p, 4 * (size_t)kwords);
    for (int bit = 64 * (int)kwords; bit >= 0; --bit) {
      for (int i = rw - 1; i > 0; --i) rem[i] = (rem[i] << 1) | (rem[i - 1] >> 31);
      rem[0] = (rem[0] << 1) | (bit == 64 * (int)kwords ? 1u : 0u);
      if (geq_words(rem, pw, rw)) {
        sub_words(rem, pw, rw);
        mu[bit / 32] |= 1u << (bit % 32);   // (bit / 32 <= kwords + 1: the quotient is at most b^(kwords + 1))
      }
    }
    return;
  }
  // R = 2^(32*nwords).  one = R mod p by doubling; r2 = R^2 mod p the same way.
  uint32_t x[W] = {1};
  for (uint32_t i = 0; i < 32 * nwords; ++i) add(x, x, x);
  memcpy(one, x, sizeof x);
  for (uint32_t i = 0; i < 32 * nwords; ++i) add(x, x, x);
  memcpy(r2, x, sizeof x);
  uint32_t inv = 1;  // Newton iteration for p^{-1} mod 2^32
  for (int i = 0; i < 5; ++i) inv *= 2 - p[0] * inv;
  n0inv = 0u - inv;
}

bool FieldHost::is_canonical(const Value& v) const {
  const size_t n = significant_bytes(v);
  if (n > 4 * (size_t)W) return false;
  uint32_t w[W] = {0};
  for (size_t i = 0; i < n; ++i) w[i / 4] |= (uint32_t)v[i] << (8 * (i % 4));
  return !geq_words(w, p, W);
}

void FieldHost::reduce(const Value& v, uint32_t out[W]) const {
  uint32_t r[W] = {0};
  const uint32_t one_[W] = {1};
  const size_t n = significant_bytes(v);
  for (size_t bit = n * 8; bit-- > 0;) {
    add(r, r, r);
    if ((v[bit / 8] >> (bit % 8)) & 1) add(r, one_, r);
  }
  memcpy(out, r, sizeof r);
}

void FieldHost::to_mont(const uint32_t in[W], uint32_t out[W]) const {
  uint32_t x[W];
  memcpy(x, in, sizeof x);
  if (!generic)
    for (uint32_t i = 0; i < 32 * nwords; ++i) add(x, x, x);
  memcpy(out, x, sizeof x);
}

// -------------------------------------------------------------- TapeBackend
void TapeBackend::set_field(const Value& modulus, uint32_t degree, bool is_boolean) {
  // PlaintextBackend::set_field checks (evaluator.rs:866-875), same strings
  FieldHost f;
  f.init(modulus, force_generic_);
  if (degree != 1) throw Error("Field should be of degree 1");
  if (field_set_) {
    // A new modulus opens a new field segment of the session (capi.cpp switch_field), which gives it a backend of its
    // own before the Evaluator gets here; a caller that drives this backend itself (zkgpu_backend_*) owns wires the
    // library cannot see, so for it the change stays refused.
    if (memcmp(f.p, field_.p, sizeof f.p) != 0)
      throw Error("GPU backend: the field characteristic changed on a backend that already holds wires (the bundled Evaluator "
                  "opens a new field segment for that, evaluator.rs:232-237; a caller-driven backend cannot)");
    is_boolean_ = is_boolean;   // the gate set may change from message to message: it only selects the Evaluator's own arms
    return;
  }
  field_ = f;
  field_set_ = true;
  is_boolean_ = is_boolean;
  modulus_ = modulus;
}

void TapeBackend::need_field() const {
  if (!field_set_) throw Error("Modulus is not initiated, used `set_field()` before calling.");
}

TapeBackend::FieldElement TapeBackend::minus_one() const {  // evaluator.rs:881-886
  need_field();
  TapeElement e;
  e.bytes = modulus_;
  size_t i = 0;
  while (i < e.bytes.size() && e.bytes[i] == 0) e.bytes[i++] = 0xff;
  if (i < e.bytes.size()) e.bytes[i] -= 1;
  return e;
}

uint32_t TapeBackend::push(uint8_t kind, uint32_t a, uint32_t b) {
  if (tape_.kind.size() >= max_ops_ || tape_.kind.size() >= 0xFFFFFFF0u)
    throw Error("GPU backend: the relation unrolls to more than " + std::to_string(max_ops_) +
                " backend operations (option max_tape_ops)");
  if (tape_.window_ops) note_level(kind, a, b);   // (may cut the tape in FRONT of this entry)
  tape_.kind.push_back(kind);
  tape_.a.push_back(a);
  tape_.b.push_back(b);
  if (kind != TK_ASSERT && kind != TK_CARRY) ++tape_.n_value_ops;
  const uint32_t h = (uint32_t)(tape_.kind.size() - 1);
  maybe_cut();
  return h;
}

// Streaming: where to cut.  A window is levelised on its own, behind everything the windows before it hold, so a cut in
// the middle of a dependency level splits that level in two (a partial level costs a launch, or a barrier and a padded
// row of the GF(2) kernel).  The recording knows the dependency depth of every entry (copies are transparent, sources have
// depth 0 -- the scheduler's own levels differ in detail, this only has to find the seams): the entry that is the first
// to reach a NEW greatest depth opens a level that nothing recorded before it belongs to -- the tape is cut in front of
// such an entry once the window is full.  A tape that is not recorded level by level gets its cut at twice the window
// (maybe_cut).  Cuts depend on the tape alone.
void TapeBackend::note_level(uint8_t kind, uint32_t a, uint32_t b) {
  uint32_t d = 0;
  switch (kind) {
    case TK_ADD: case TK_MUL: case TK_AND: case TK_XOR: d = std::max(depth_[a], depth_[b]) + 1; break;
    case TK_ADDC: case TK_MULC: case TK_NOT: case TK_ASSERT: case TK_NZ: d = depth_[a] + 1; break;
    case TK_COPY: d = depth_[a]; break;
    default: break;   // constant, instance, witness, carried value
  }
  if (d > top_depth_) {
    top_depth_ = d;
    if (tape_.ladder_open == kNoWire) {
      const uint32_t last = tape_.cuts.empty() ? 0 : tape_.cuts.back();
      if (tape_.size() - last >= tape_.window_ops) {
        tape_.cuts.push_back((uint32_t)tape_.size());
        if (cut_hook_) cut_hook_(cut_arg_);
      }
    }
  }
  depth_.push_back(d);
}

uint32_t TapeBackend::arith(uint8_t kind, uint32_t a, uint32_t b) {
  need_field();
  return push(kind, a, b);  // for p == 2 the scheduler lowers add/mul to xor/and on bit-packed wires
}

uint32_t TapeBackend::bitwise(uint8_t kind, uint32_t a, uint32_t b) {
  need_field();
  // PlaintextBackend applies & ^ to the integers and then `% m`, and `not` is `is_zero ? 1 : 0`
  // (evaluator.rs:924-938); for p == 2 that is the bit-packed path, for an odd p the arithmetic kernels do the
  // same on the canonical values (bit_operation, device/replay_kernels.hpp; fp_is_zero_indicator, device/fp_mont.hpp).
  return push(kind, a, b);
}

uint32_t TapeBackend::intern(const Value& bytes) {
  auto it = const_index_.find(bytes);
  if (it != const_index_.end()) return it->second;
  const uint32_t idx = (uint32_t)tape_.consts.size();
  tape_.consts.push_back(bytes);
  const_index_.emplace(bytes, idx);
  return idx;
}

uint32_t TapeBackend::h_constant(FieldElement val) {
  need_field();
  if (val.kind != TapeElement::LITERAL) throw Error("GPU backend: constant() needs literal bytes");
  // constant() stores the integer unreduced in the reference (evaluator.rs:896-898).  Here it is reduced, which is the
  // same thing wherever the value first meets an arithmetic gate; b = 1 marks a value >= p so that the scheduler can
  // refuse the cases where it is not (schedule.cpp mark_strict_sources).
  return push(TK_CONST, intern(val.bytes), field_.is_canonical(val.bytes) ? 0u : 1u);
}

uint32_t TapeBackend::h_add_constant(uint32_t x, FieldElement c) {
  need_field();
  if (c.kind != TapeElement::LITERAL) throw Error("GPU backend: add_constant() needs literal bytes");
  return push(TK_ADDC, x, intern(c.bytes));  // (a + c) % m: c may be reduced first
}

uint32_t TapeBackend::h_mul_constant(uint32_t x, FieldElement c) {
  need_field();
  if (c.kind != TapeElement::LITERAL) throw Error("GPU backend: mul_constant() needs literal bytes");
  return push(TK_MULC, x, intern(c.bytes));
}

void TapeBackend::h_assert_zero(uint32_t w) {
  need_field();
  const uint32_t seq = assert_base_ + (uint32_t)tape_.assert_op.size();   // global over the field segments of the session
  // the bookkeeping goes first: push() may close a window and hand the tape over
  tape_.assert_op.push_back((uint32_t)tape_.size());
  tape_.assert_wire.push_back(pending_assert_wire_);
  push(TK_ASSERT, w, seq);
}

TapeBackend::FieldElement TapeBackend::instance_ref(uint32_t position) {
  TapeElement e;
  e.kind = TapeElement::INSTANCE_REF;
  e.position = position;
  return e;
}
TapeBackend::FieldElement TapeBackend::witness_ref(uint32_t position) {
  TapeElement e;
  e.kind = TapeElement::WITNESS_REF;
  e.position = position;
  return e;
}
TapeBackend::FieldElement TapeBackend::import_instance(const Value& v) {
  lane0_instances_.push_back(v);
  return instance_ref((uint32_t)lane0_instances_.size() - 1);
}
TapeBackend::FieldElement TapeBackend::import_witness(const Value& v) {
  lane0_witnesses_.push_back(v);
  return witness_ref((uint32_t)lane0_witnesses_.size() - 1);
}

uint32_t TapeBackend::h_instance(FieldElement val) {
  need_field();
  if (val.kind != TapeElement::INSTANCE_REF) throw Error("GPU backend: instance() needs a stream position");
  if (val.position + 1 > tape_.n_instance) tape_.n_instance = val.position + 1;
  return push(TK_INSTANCE, val.position, 0);
}

uint32_t TapeBackend::h_carry(uint32_t index) {
  need_field();
  if (index + 1 > tape_.n_carry) tape_.n_carry = index + 1;
  return push(TK_CARRY, index, 0);
}

uint32_t TapeBackend::h_input_at(uint8_t kind, uint32_t position) {
  need_field();
  if (kind != TK_INSTANCE && kind != TK_WITNESS) throw Error("GPU backend: h_input_at() names an input stream");
  uint32_t& n = kind == TK_INSTANCE ? tape_.n_instance : tape_.n_witness;
  if (position + 1 > n) n = position + 1;
  return push(kind, position, 0);
}

uint32_t TapeBackend::h_witness(const FieldElement* val) {
  need_field();
  // PlaintextBackend panics on a missing witness (evaluator.rs:944-946)
  if (!val) throw Panic("Missing witness value for PlaintextBackend");
  if (val->kind != TapeElement::WITNESS_REF) throw Error("GPU backend: witness() needs a stream position");
  if (val->position + 1 > tape_.n_witness) tape_.n_witness = val->position + 1;
  return push(TK_WITNESS, val->position, 0);
}

}  // namespace zki
