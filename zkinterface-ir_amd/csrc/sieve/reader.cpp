#include "reader.hpp"

#include <dirent.h>
#include <string.h>
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <fstream>

namespace zki {
namespace {

// vtable byte offsets (VT_* in rust/src/sieve_ir_generated.rs; table in SURVEY.md 5.9)
namespace vt {
enum : uint16_t {
  ROOT_TYPE = 4, ROOT_MESSAGE = 6,
  HEADER_VERSION = 4, HEADER_FIELD_CHARACTERISTIC = 6, HEADER_FIELD_DEGREE = 8,
  RELATION_HEADER = 4, RELATION_GATESET = 6, RELATION_FEATURES = 8, RELATION_FUNCTIONS = 10, RELATION_DIRECTIVES = 12,
  INPUTS_HEADER = 4, INPUTS_VALUES = 6,  // Instance.common_inputs / Witness.short_witness
  WIRE_ID = 4, VALUE_VALUE = 4,
  RANGE_FIRST = 4, RANGE_LAST = 6,
  ELEMENT_TYPE = 4, ELEMENT = 6,  // every (type, value) union pair in the schema uses slots 4/6
  LIST_ELEMENTS = 4,
  GATE_OUTPUT = 4, GATE_IN0 = 6, GATE_IN1 = 8,
  CONSTANT_VALUE = 6, CONSTGATE_VALUE = 8,
  FREE_FIRST = 4, FREE_LAST = 6,
  FUNCTION_NAME = 4, FUNCTION_OUTPUT_COUNT = 6, FUNCTION_INPUT_COUNT = 8, FUNCTION_INSTANCE_COUNT = 10,
  FUNCTION_WITNESS_COUNT = 12, FUNCTION_BODY = 14,
  CALL_NAME = 4, CALL_OUTPUTS = 6, CALL_INPUTS = 8,
  ANONCALL_OUTPUTS = 4, ANONCALL_INNER = 6,
  ABSTRACT_CALL_NAME = 4, ABSTRACT_CALL_INPUTS = 6,
  ABSTRACT_ANON_INPUTS = 4, ABSTRACT_ANON_INSTANCE_COUNT = 6, ABSTRACT_ANON_WITNESS_COUNT = 8, ABSTRACT_ANON_SUBCIRCUIT = 10,
  SWITCH_CONDITION = 4, SWITCH_OUTPUTS = 6, SWITCH_CASES = 8, SWITCH_BRANCHES = 10,
  ITER_LEFT = 4, ITER_RIGHT = 6, ITER_CONST_VALUE = 4, ITER_NAME = 4, ITER_DIV_NUMER = 4, ITER_DIV_DENOM = 6,
  ITER_INVOKE_NAME = 4, ITER_INVOKE_OUTPUTS = 6, ITER_INVOKE_INPUTS = 8,
  ITER_ANON_OUTPUTS = 4, ITER_ANON_INPUTS = 6, ITER_ANON_INSTANCE_COUNT = 8, ITER_ANON_WITNESS_COUNT = 10, ITER_ANON_BODY = 12,
  FOR_OUTPUTS = 4, FOR_ITERATOR = 6, FOR_FIRST = 8, FOR_LAST = 10, FOR_BODY_TYPE = 12, FOR_BODY = 14,
};
}  // namespace vt

// Bounds-checked view over one FlatBuffer (the bytes after the size prefix).
class Bytes {
 public:
  Bytes(const uint8_t* p, size_t n) : p_(p), n_(n) {}
  template <typename T>
  T read(size_t at) const {
    if (at > n_ || sizeof(T) > n_ - at) throw Panic("flatbuffer: read past end of message");
    T v;
    memcpy(&v, p_ + at, sizeof(T));
    return v;
  }
  const uint8_t* span(size_t at, size_t len) const {
    if (at > n_ || len > n_ - at) throw Panic("flatbuffer: vector past end of message");
    return p_ + at;
  }
  size_t size() const { return n_; }

 private:
  const uint8_t* p_;
  size_t n_;
};

class Vec;

class Tab {
 public:
  Tab() = default;
  Tab(const Bytes* b, size_t at) : b_(b), at_(at) {}
  explicit operator bool() const { return b_ != nullptr; }

  template <typename T>
  T scalar(uint16_t slot) const {  // absent field -> schema default 0
    const size_t f = locate(slot);
    return f ? b_->read<T>(f) : T(0);
  }
  Tab child(uint16_t slot) const {
    const size_t f = locate(slot);
    return f ? Tab(b_, f + b_->read<uint32_t>(f)) : Tab();
  }
  inline Vec vec(uint16_t slot) const;
  bool text(uint16_t slot, std::string* out) const;
  bool bytes(uint16_t slot, Value* out) const;
  const Bytes* buf() const { return b_; }

 private:
  size_t locate(uint16_t slot) const {
    if (!b_) throw Panic("called `Option::unwrap()` on a `None` value");
    const int32_t back = b_->read<int32_t>(at_);
    const size_t vtab = (size_t)((int64_t)at_ - back);
    const uint16_t vt_len = b_->read<uint16_t>(vtab);
    if ((size_t)slot + 2 > vt_len) return 0;
    const uint16_t rel = b_->read<uint16_t>(vtab + slot);
    return rel ? at_ + rel : 0;
  }
  const Bytes* b_ = nullptr;
  size_t at_ = 0;
};

class Vec {  // vector of offsets to tables, or raw bytes
 public:
  Vec() = default;
  Vec(const Bytes* b, size_t at) : b_(b), at_(at), n_(b->read<uint32_t>(at)) {
    // a length that cannot fit in the message is corruption: refuse before anything is reserved
    if ((size_t)n_ > b->size() - std::min(b->size(), at + 4)) throw Panic("flatbuffer: vector length exceeds the message");
  }
  explicit operator bool() const { return b_ != nullptr; }
  uint32_t size() const { return n_; }
  Tab table(uint32_t i) const {
    const size_t e = at_ + 4 + 4 * (size_t)i;
    return Tab(b_, e + b_->read<uint32_t>(e));
  }
  const uint8_t* raw() const { return b_->span(at_ + 4, n_); }

 private:
  const Bytes* b_ = nullptr;
  size_t at_ = 0;
  uint32_t n_ = 0;
};

inline Vec Tab::vec(uint16_t slot) const {
  const size_t f = locate(slot);
  return f ? Vec(b_, f + b_->read<uint32_t>(f)) : Vec();
}
bool Tab::text(uint16_t slot, std::string* out) const {
  Vec v = vec(slot);
  if (!v) return false;
  out->assign(reinterpret_cast<const char*>(v.raw()), v.size());
  return true;
}
bool Tab::bytes(uint16_t slot, Value* out) const {
  Vec v = vec(slot);
  if (!v) return false;
  const uint8_t* p = v.raw();
  out->assign(p, p + v.size());
  return true;
}

// Nested tables (iterator expressions, subcircuits) are walked recursively; a zero or cyclic offset in a
// corrupt message must not recurse without end.
constexpr int kMaxNesting = 256;
struct NestingGuard {
  explicit NestingGuard(int* d) : depth(d) {
    if (++*depth > kMaxNesting) {
      --*depth;
      throw Panic("flatbuffer: tables nested deeper than 256 levels");
    }
  }
  ~NestingGuard() { --*depth; }
  int* depth;
};
thread_local int g_nesting = 0;

WireId need_wire(const Tab& t, uint16_t slot, const char* what) {
  Tab w = t.child(slot);
  if (!w) throw Error(what);
  return w.scalar<uint64_t>(vt::WIRE_ID);
}

Value decode_value(const Tab& t) {  // structs/value.rs:14-16
  Value v;
  if (!t.bytes(vt::VALUE_VALUE, &v)) throw Error("Missing value");
  return v;
}
std::vector<Value> decode_values(const Vec& v) {  // structs/value.rs:19-29
  std::vector<Value> out;
  out.reserve(v.size());
  for (uint32_t i = 0; i < v.size(); ++i) out.push_back(decode_value(v.table(i)));
  return out;
}

WireList decode_wirelist(const Tab& t) {  // structs/wire.rs:90-157
  Vec els = t.vec(vt::LIST_ELEMENTS);
  if (!els) throw Error("Missing wire list");
  WireList out;
  out.reserve(els.size());
  for (uint32_t i = 0; i < els.size(); ++i) {
    Tab el = els.table(i);
    const uint8_t ty = el.scalar<uint8_t>(vt::ELEMENT_TYPE);
    Tab inner = el.child(vt::ELEMENT);
    WireRange r;
    if (ty == 1) {
      r.first = r.last = inner.scalar<uint64_t>(vt::WIRE_ID);
    } else if (ty == 2) {
      r.range = true;
      r.first = need_wire(inner, vt::RANGE_FIRST, "Missing first value of range");
      r.last = need_wire(inner, vt::RANGE_LAST, "Missing last value of range");
    } else {
      throw Error("Unknown type in WireListElement");
    }
    out.push_back(r);
  }
  return out;
}
WireList need_wirelist(const Tab& t, uint16_t slot, const char* what) {
  Tab l = t.child(slot);
  if (!l) throw Error(what);
  return decode_wirelist(l);
}

IterExpr decode_iterexpr(const Tab& t) {  // structs/iterators.rs:35-114
  NestingGuard guard(&g_nesting);
  IterExpr e;
  const uint8_t ty = t.scalar<uint8_t>(vt::ELEMENT_TYPE);
  Tab v = t.child(vt::ELEMENT);
  auto operand = [&](uint16_t slot, const char* what) {
    Tab c = v.child(slot);
    if (!c) throw Error(what);
    e.args.push_back(decode_iterexpr(c));
  };
  switch (ty) {
    case IterExpr::CONST:
      e.op = IterExpr::CONST;
      e.value = v.scalar<uint64_t>(vt::ITER_CONST_VALUE);
      break;
    case IterExpr::NAME:
      e.op = IterExpr::NAME;
      if (!v.text(vt::ITER_NAME, &e.name)) throw Error("IterExpr: No name given");
      break;
    case IterExpr::ADD:
    case IterExpr::SUB:
    case IterExpr::MUL:
      e.op = (IterExpr::Op)ty;
      operand(vt::ITER_LEFT, "Missing left operand");
      operand(vt::ITER_RIGHT, "Missing right operand");
      break;
    case IterExpr::DIV_CONST:
      e.op = IterExpr::DIV_CONST;
      operand(vt::ITER_DIV_NUMER, "Missing numerator");
      e.value = v.scalar<uint64_t>(vt::ITER_DIV_DENOM);
      break;
    default: throw Error("Unknown Iterator Expression type");
  }
  return e;
}
IterExprList decode_iterlist(const Tab& t) {  // structs/iterators.rs:244-270,314-326
  Vec els = t.vec(vt::LIST_ELEMENTS);
  if (!els) throw Error("Missing wire list");
  IterExprList out;
  for (uint32_t i = 0; i < els.size(); ++i) {
    Tab el = els.table(i);
    const uint8_t ty = el.scalar<uint8_t>(vt::ELEMENT_TYPE);
    Tab inner = el.child(vt::ELEMENT);
    IterExprRange r;
    if (ty == 1) {
      r.first = decode_iterexpr(inner);
    } else if (ty == 2) {
      r.range = true;
      Tab f = inner.child(vt::RANGE_FIRST), l = inner.child(vt::RANGE_LAST);
      if (!f) throw Error("Missing first value of range");
      r.first = decode_iterexpr(f);
      if (!l) throw Error("Missing last value of range");
      r.last = decode_iterexpr(l);
    } else {
      throw Error("Unknown type in IterExprWireListElement");
    }
    out.push_back(std::move(r));
  }
  return out;
}
IterExprList need_iterlist(const Tab& t, uint16_t slot, const char* what) {
  Tab l = t.child(slot);
  if (!l) throw Error(what);
  return decode_iterlist(l);
}

std::shared_ptr<Subcircuit> decode_gates(const Vec& v);

CaseInvoke decode_case(const Tab& t) {  // structs/function.rs:132-172
  CaseInvoke c;
  const uint8_t ty = t.scalar<uint8_t>(vt::ELEMENT_TYPE);
  Tab inv = t.child(vt::ELEMENT);
  if (ty == 1) {
    if (!inv.text(vt::ABSTRACT_CALL_NAME, &c.name)) throw Error("Missing function name.");
    c.input_wires = need_wirelist(inv, vt::ABSTRACT_CALL_INPUTS, "Missing inputs");
  } else if (ty == 2) {
    c.anonymous = true;
    Vec body = inv.vec(vt::ABSTRACT_ANON_SUBCIRCUIT);
    if (!body) throw Error("Missing implementation");
    c.subcircuit = decode_gates(body);
    c.input_wires = need_wirelist(inv, vt::ABSTRACT_ANON_INPUTS, "Missing inputs");
    c.instance_count = inv.scalar<uint64_t>(vt::ABSTRACT_ANON_INSTANCE_COUNT);
    c.witness_count = inv.scalar<uint64_t>(vt::ABSTRACT_ANON_WITNESS_COUNT);
  } else {
    throw Error("No directive type");
  }
  return c;
}

Gate decode_gate(const Tab& directive) {  // structs/gates.rs:60-259
  NestingGuard guard(&g_nesting);
  Gate g;
  const uint8_t ty = directive.scalar<uint8_t>(vt::ELEMENT_TYPE);
  if (ty == 0 || ty > (uint8_t)GateKind::For) throw Error("No gate type");
  const Tab t = directive.child(vt::ELEMENT);
  g.kind = (GateKind)ty;
  switch (g.kind) {
    case GateKind::Constant:
      g.out = need_wire(t, vt::GATE_OUTPUT, "Missing output");
      g.ext = std::make_shared<GateExt>();
      if (!t.bytes(vt::CONSTANT_VALUE, &g.ext->constant)) throw Error("Missing constant");
      break;
    case GateKind::AssertZero: g.in0 = need_wire(t, vt::GATE_OUTPUT, "Missing input"); break;
    case GateKind::Copy:
    case GateKind::Not:
      g.out = need_wire(t, vt::GATE_OUTPUT, "Missing output");
      g.in0 = need_wire(t, vt::GATE_IN0, "Missing input");
      break;
    case GateKind::Add:
    case GateKind::Mul:
    case GateKind::And:
    case GateKind::Xor:
      g.out = need_wire(t, vt::GATE_OUTPUT, "Missing output");
      g.in0 = need_wire(t, vt::GATE_IN0, "Missing left input");
      g.in1 = need_wire(t, vt::GATE_IN1, "Missing right input");
      break;
    case GateKind::AddConstant:
    case GateKind::MulConstant:
      g.out = need_wire(t, vt::GATE_OUTPUT, "Missing output");
      g.in0 = need_wire(t, vt::GATE_IN0, "Missing input");
      g.ext = std::make_shared<GateExt>();
      if (!t.bytes(vt::CONSTGATE_VALUE, &g.ext->constant)) throw Error("Missing constant");
      break;
    case GateKind::Instance:
    case GateKind::Witness: g.out = need_wire(t, vt::GATE_OUTPUT, "Missing output"); break;
    case GateKind::Free: {
      g.in0 = need_wire(t, vt::FREE_FIRST, "Missing first wire");
      Tab last = t.child(vt::FREE_LAST);
      g.has_last = (bool)last;
      if (last) g.in1 = last.scalar<uint64_t>(vt::WIRE_ID);
      break;
    }
    case GateKind::Call:
      g.ext = std::make_shared<GateExt>();
      if (!t.text(vt::CALL_NAME, &g.ext->name)) throw Error("Missing function name.");
      g.ext->output_wires = need_wirelist(t, vt::CALL_OUTPUTS, "Missing outputs");
      g.ext->input_wires = need_wirelist(t, vt::CALL_INPUTS, "Missing inputs");
      break;
    case GateKind::AnonCall: {
      g.ext = std::make_shared<GateExt>();
      Tab inner = t.child(vt::ANONCALL_INNER);
      if (!inner) throw Error("Missing inner AbstractAnonCall");
      g.ext->output_wires = need_wirelist(t, vt::ANONCALL_OUTPUTS, "Missing output wires");
      g.ext->input_wires = need_wirelist(inner, vt::ABSTRACT_ANON_INPUTS, "Missing input wires");
      g.ext->instance_count = inner.scalar<uint64_t>(vt::ABSTRACT_ANON_INSTANCE_COUNT);
      g.ext->witness_count = inner.scalar<uint64_t>(vt::ABSTRACT_ANON_WITNESS_COUNT);
      Vec body = inner.vec(vt::ABSTRACT_ANON_SUBCIRCUIT);
      if (!body) throw Error("Missing subcircuit");
      g.ext->subcircuit = decode_gates(body);
      break;
    }
    case GateKind::Switch: {
      g.ext = std::make_shared<GateExt>();
      Vec cases = t.vec(vt::SWITCH_CASES);
      if (!cases) throw Error("Missing cases values");
      g.ext->cases = decode_values(cases);
      g.in0 = need_wire(t, vt::SWITCH_CONDITION, "Missing condition wire.");
      g.ext->output_wires = need_wirelist(t, vt::SWITCH_OUTPUTS, "Missing output wires");
      Vec branches = t.vec(vt::SWITCH_BRANCHES);
      if (!branches) throw Error("Missing branches");
      for (uint32_t i = 0; i < branches.size(); ++i) g.ext->branches.push_back(decode_case(branches.table(i)));
      break;
    }
    case GateKind::For: {
      g.ext = std::make_shared<GateExt>();
      g.ext->output_wires = need_wirelist(t, vt::FOR_OUTPUTS, "missing output list");
      const uint8_t bty = t.scalar<uint8_t>(vt::FOR_BODY_TYPE);
      Tab body = t.child(vt::FOR_BODY);
      ForLoopBody& fb = g.ext->body;
      if (bty == 1) {
        if (!body.text(vt::ITER_INVOKE_NAME, &fb.name)) throw Error("Missing function in function name");
        fb.outputs = need_iterlist(body, vt::ITER_INVOKE_OUTPUTS, "missing output list");
        fb.inputs = need_iterlist(body, vt::ITER_INVOKE_INPUTS, "missing input list");
      } else if (bty == 2) {
        fb.anonymous = true;
        fb.outputs = need_iterlist(body, vt::ITER_ANON_OUTPUTS, "missing output list");
        fb.inputs = need_iterlist(body, vt::ITER_ANON_INPUTS, "missing input list");
        fb.instance_count = body.scalar<uint64_t>(vt::ITER_ANON_INSTANCE_COUNT);
        fb.witness_count = body.scalar<uint64_t>(vt::ITER_ANON_WITNESS_COUNT);
        Vec sub = body.vec(vt::ITER_ANON_BODY);
        if (!sub) throw Error("Missing body");
        fb.subcircuit = decode_gates(sub);
      } else {
        throw Error("Unknown body type");
      }
      if (!t.text(vt::FOR_ITERATOR, &g.ext->name)) throw Error("Missing iterator name");
      g.ext->first = t.scalar<uint64_t>(vt::FOR_FIRST);
      g.ext->last = t.scalar<uint64_t>(vt::FOR_LAST);
      break;
    }
    default: throw Error("No gate type");
  }
  return g;
}

std::shared_ptr<Subcircuit> decode_gates(const Vec& v) {  // structs/gates.rs:682-691
  auto out = std::make_shared<Subcircuit>();
  out->reserve(v.size());
  for (uint32_t i = 0; i < v.size(); ++i) out->push_back(decode_gate(v.table(i)));
  return out;
}

Header decode_header(const Tab& h) {  // structs/header.rs:37-56
  if (!h) throw Error("Missing header");
  Header out;
  if (!h.text(vt::HEADER_VERSION, &out.version)) throw Error("Missing version");
  Tab fc = h.child(vt::HEADER_FIELD_CHARACTERISTIC);
  if (!fc) throw Error("Missing field characteristic");
  out.field_characteristic = decode_value(fc);
  out.field_degree = h.scalar<uint32_t>(vt::HEADER_FIELD_DEGREE);
  return out;
}

std::string without_spaces(const std::string& s) {
  std::string r;
  for (char c : s)
    if (c != ' ') r.push_back(c);
  return r;
}
template <typename F>
void for_each_token(const std::string& s, F&& f) {  // str::split(',')
  size_t b = 0;
  while (true) {
    const size_t e = s.find(',', b);
    if (f(s.substr(b, e == std::string::npos ? std::string::npos : e - b))) return;
    if (e == std::string::npos) return;
    b = e + 1;
  }
}

bool has_sieve_extension(const std::string& path) {  // source.rs:161-163
  const size_t slash = path.find_last_of('/');
  const std::string name = slash == std::string::npos ? path : path.substr(slash + 1);
  const size_t dot = name.find_last_of('.');
  return dot != std::string::npos && dot != 0 && name.substr(dot + 1) == "sieve";
}
std::string file_name(const std::string& path) {
  const size_t slash = path.find_last_of('/');
  return slash == std::string::npos ? path : path.substr(slash + 1);
}

}  // namespace

uint16_t parse_gate_set(const std::string& gateset) {
  uint16_t ret = 0;
  for_each_token(gateset, [&](const std::string& raw) {
    const std::string s = without_spaces(raw);
    if (s == "arithmetic") { ret = mask::ARITH; return true; }  // short-circuits (relation.rs:149)
    if (s == "boolean") { ret = mask::BOOL; return true; }
    if (s == "@add") ret |= mask::ADD;
    else if (s == "@addc") ret |= mask::ADDC;
    else if (s == "@mul") ret |= mask::MUL;
    else if (s == "@mulc") ret |= mask::MULC;
    else if (s == "@xor") ret |= mask::XOR;
    else if (s == "@not") ret |= mask::NOT;
    else if (s == "@and") ret |= mask::AND;
    else if (!s.empty()) throw Error("Unable to parse the following gateset: " + gateset);
    return false;
  });
  return ret;
}

uint16_t parse_feature_toggle(const std::string& features) {
  uint16_t ret = 0;
  for_each_token(features, [&](const std::string& raw) {
    const std::string s = without_spaces(raw);
    if (s == "simple") { ret = mask::SIMPLE; return true; }
    if (s == "@function") ret |= mask::FUNCTION;
    else if (s == "@for") ret |= mask::FOR;
    else if (s == "@switch") ret |= mask::SWITCH;
    else if (!s.empty()) throw Error("Unable to parse following feature toggles " + raw);
    return false;
  });
  return ret;
}

Message read_message(const uint8_t* data, size_t len) {
  if (len < 8) throw Panic("flatbuffer: message shorter than its header");
  Bytes body(data + 4, len - 4);
  Tab root(&body, body.read<uint32_t>(0));
  const uint8_t ty = root.scalar<uint8_t>(vt::ROOT_TYPE);
  Tab m = root.child(vt::ROOT_MESSAGE);
  Message msg;
  switch (ty) {
    case Message::IsRelation: {  // structs/relation.rs:47-72
      msg.kind = Message::IsRelation;
      Relation& r = msg.relation;
      Vec directives = m.vec(vt::RELATION_DIRECTIVES);
      if (!directives) throw Error("Missing directives");
      if (Vec fns = m.vec(vt::RELATION_FUNCTIONS)) {
        for (uint32_t i = 0; i < fns.size(); ++i) {  // structs/function.rs:29-46
          Tab ft = fns.table(i);
          Function f;
          Vec body = ft.vec(vt::FUNCTION_BODY);
          if (!body) throw Error("Missing reference implementation");
          if (!ft.text(vt::FUNCTION_NAME, &f.name)) throw Error("Missing name");
          f.output_count = ft.scalar<uint64_t>(vt::FUNCTION_OUTPUT_COUNT);
          f.input_count = ft.scalar<uint64_t>(vt::FUNCTION_INPUT_COUNT);
          f.instance_count = ft.scalar<uint64_t>(vt::FUNCTION_INSTANCE_COUNT);
          f.witness_count = ft.scalar<uint64_t>(vt::FUNCTION_WITNESS_COUNT);
          f.body = decode_gates(body);
          r.functions.push_back(std::move(f));
        }
      }
      r.header = decode_header(m.child(vt::RELATION_HEADER));
      std::string s;
      if (!m.text(vt::RELATION_GATESET, &s)) throw Error("Missing gateset description");
      r.gate_mask = parse_gate_set(s);
      if (!m.text(vt::RELATION_FEATURES, &s)) throw Error("Missing feature toggles");
      r.feat_mask = parse_feature_toggle(s);
      r.gates = std::move(*decode_gates(directives));
      break;
    }
    case Message::IsInstance: {  // structs/instance.rs:18-31
      msg.kind = Message::IsInstance;
      msg.instance.header = decode_header(m.child(vt::INPUTS_HEADER));
      Vec v = m.vec(vt::INPUTS_VALUES);
      if (!v) throw Error("Missing common_input");
      msg.instance.common_inputs = decode_values(v);
      break;
    }
    case Message::IsWitness: {  // structs/witness.rs:18-31
      msg.kind = Message::IsWitness;
      msg.witness.header = decode_header(m.child(vt::INPUTS_HEADER));
      Vec v = m.vec(vt::INPUTS_VALUES);
      if (!v) throw Error("Missing short_witness");
      msg.witness.short_witness = decode_values(v);
      break;
    }
    default: throw Error("Invalid message type");
  }
  return msg;
}

namespace {
// The reference materialises every expanded list (wire.rs:178-203) and would exhaust memory on an
// absurd range; the host refuses instead.
constexpr uint64_t kMaxExpandedWires = 1ull << 28;
void check_expansion(uint64_t have, uint64_t first, uint64_t last) {
  if (last - first >= kMaxExpandedWires || have + (last - first) >= kMaxExpandedWires)
    throw Error("wire list expands to more than 2^28 wires");
}
}  // namespace

std::vector<WireId> expand_wirelist(const WireList& list) {
  std::vector<WireId> out;
  for (const WireRange& r : list) {
    if (!r.range) {
      out.push_back(r.first);
      continue;
    }
    if (r.last <= r.first)
      throw Error("In WireRange, last WireId (" + std::to_string(r.last) +
                  ") must be strictly greater than first WireId (" + std::to_string(r.first) + ").");
    check_expansion(out.size(), r.first, r.last);
    for (WireId w = r.first; w <= r.last; ++w) {
      out.push_back(w);
      if (w == UINT64_MAX) break;
    }
  }
  return out;
}

const uint64_t* IteratorScope::find(const std::string& name) const {
  for (const auto& kv : vars)
    if (kv.first == name) return &kv.second;
  return nullptr;
}
void IteratorScope::insert(const std::string& name, uint64_t v) {
  for (auto& kv : vars)
    if (kv.first == name) { kv.second = v; return; }
  vars.emplace_back(name, v);
}
void IteratorScope::remove(const std::string& name) {
  for (size_t i = 0; i < vars.size(); ++i)
    if (vars[i].first == name) { vars.erase(vars.begin() + i); return; }
}

namespace {
uint64_t eval_iterexpr(const IterExpr& e, const IteratorScope& known) {  // structs/iterators.rs:349-373
  switch (e.op) {
    case IterExpr::CONST: return e.value;
    case IterExpr::NAME: {
      const uint64_t* v = known.find(e.name);
      if (!v) throw Panic("Unknown iterator name " + e.name);  // unwrap_or_else(panic!) at :399-400
      return *v;
    }
    case IterExpr::ADD: return eval_iterexpr(e.args[0], known) + eval_iterexpr(e.args[1], known);
    case IterExpr::SUB: return eval_iterexpr(e.args[0], known) - eval_iterexpr(e.args[1], known);
    case IterExpr::MUL: return eval_iterexpr(e.args[0], known) * eval_iterexpr(e.args[1], known);
    case IterExpr::DIV_CONST: {
      const uint64_t n = eval_iterexpr(e.args[0], known);
      if (e.value == 0) throw Panic("attempt to divide by zero");
      return n / e.value;
    }
  }
  throw Panic("corrupt iterator expression");
}
}  // namespace

std::vector<WireId> evaluate_iterexpr_list(const IterExprList& list, const IteratorScope& known) {
  std::vector<WireId> out;
  for (const IterExprRange& r : list) {
    const uint64_t a = eval_iterexpr(r.first, known);
    if (!r.range) {
      out.push_back(a);
      continue;
    }
    const uint64_t b = eval_iterexpr(r.last, known);
    if (a <= b) check_expansion(out.size(), a, b);
    for (uint64_t w = a; w <= b; ++w) {  // first..=last: empty when first > last
      out.push_back(w);
      if (w == UINT64_MAX) break;
    }
  }
  return out;
}

std::vector<std::pair<size_t, size_t>> split_messages(const uint8_t* data, size_t len) {
  std::vector<std::pair<size_t, size_t>> out;
  size_t at = 0;
  while (len - at >= 4) {
    uint32_t body;
    memcpy(&body, data + at, 4);
    if (body == 0) break;                       // explicit end marker
    if ((size_t)body > len - at - 4) break;     // read_exact would fail: treated as end of stream
    out.emplace_back(at, 4 + (size_t)body);
    at += 4 + (size_t)body;
  }
  return out;
}

Source Source::from_directory(const std::string& path) { return from_dirs_and_files({path}); }

Source Source::from_dirs_and_files(const std::vector<std::string>& paths) {  // source.rs:64-67,165-193
  std::vector<std::string> all;
  for (const std::string& p : paths) {
    if (has_sieve_extension(p)) {
      all.push_back(p);
    } else if (p == "-") {
      throw Error("stdin sources are not supported by this host");
    } else {
      DIR* d = opendir(p.c_str());
      if (!d) throw Error("cannot read directory " + p);
      while (dirent* e = readdir(d)) {
        std::string child = p + (p.empty() || p.back() == '/' ? "" : "/") + e->d_name;
        if (has_sieve_extension(child)) all.push_back(child);
      }
      closedir(d);
    }
  }
  return from_filenames(std::move(all));
}

Source Source::from_filenames(std::vector<std::string> paths) {  // source.rs:69-89
  std::sort(paths.begin(), paths.end());
  auto rank = [](const std::string& p) {
    const std::string name = file_name(p);
    if (name.find("instance") != std::string::npos) return 0;
    if (name.find("witness") != std::string::npos) return 1;
    if (name.find("relation") != std::string::npos) return 3;
    return 4;
  };
  std::stable_sort(paths.begin(), paths.end(),
                   [&](const std::string& a, const std::string& b) { return rank(a) < rank(b); });
  Source s;
  s.files_ = std::move(paths);
  s.from_files_ = true;
  return s;
}

Source Source::from_buffers(std::vector<std::vector<uint8_t>> buffers) {
  Source s;
  s.buffers_ = std::move(buffers);
  return s;
}

void Source::for_each_buffer(const std::function<void(const uint8_t*, size_t)>& fn) const {
  auto stream = [&](const std::vector<uint8_t>& bytes) {
    for (const auto& m : split_messages(bytes.data(), bytes.size())) fn(bytes.data() + m.first, m.second);
  };
  if (from_files_) {
    for (const std::string& path : files_) {
      if (print_filenames) fprintf(stderr, "Reading %s\n", path.c_str());
      std::ifstream f(path, std::ios::binary);
      if (!f) {
        fprintf(stderr, "Warning: failed to open file %s\n", path.c_str());
        continue;
      }
      std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      stream(bytes);
    }
  } else {
    for (const auto& b : buffers_) stream(b);
  }
}

void Source::for_each_message(const std::function<void(Message&&)>& fn) const {
  for_each_buffer([&](const uint8_t* p, size_t n) { fn(read_message(p, n)); });
}

Messages Source::read_all_messages() const {
  Messages all;
  for_each_message([&](Message&& m) { all.push_message(std::move(m)); });
  return all;
}

}  // namespace zki
