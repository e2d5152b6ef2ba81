// TEST INFRASTRUCTURE ONLY (oracle/): never linked into, imported by or called
// from the product path.
//
// Minimal read-only FlatBuffers walker for the SIEVE IR schema
// (/root/reference/sieve_ir.fbs) producing owned structs with the same field
// names as the reference's rust/src/structs/*.rs.  Vtable slot numbers are the
// VT_* constants of rust/src/sieve_ir_generated.rs (listed in SURVEY.md 5.9).
// Decode errors carry the same strings as the reference's TryFrom impls
// (e.g. rust/src/structs/gates.rs:60-259).
#pragma once
#include <stdint.h>
#include <string.h>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace zko {

typedef uint64_t WireId;
typedef std::vector<uint8_t> Value;

struct Err : std::runtime_error {
  explicit Err(const std::string& s) : std::runtime_error(s) {}
};

// ---------------- owned structs (rust/src/structs) ----------------
struct WireListElement {  // structs/wire.rs:10-13
  bool is_range;
  WireId first, last;
};
typedef std::vector<WireListElement> WireList;

struct IterExprWireNumber {  // structs/iterators.rs:17-30
  enum Kind { Const = 1, Name = 2, Add = 3, Sub = 4, Mul = 5, DivConst = 6 } kind;
  uint64_t value = 0;  // Const value / DivConst denominator
  std::string name;
  std::shared_ptr<IterExprWireNumber> left, right;  // DivConst: left = numer
};
struct IterExprListElement {  // structs/iterators.rs: Single / Range
  bool is_range;
  IterExprWireNumber first, last;
};
typedef std::vector<IterExprListElement> IterExprList;

struct Gate;
struct CaseInvoke {  // structs/function.rs:120-131
  bool is_anon;
  std::string name;
  WireList input_wires;
  uint64_t instance_count = 0, witness_count = 0;
  std::vector<Gate> subcircuit;
};
struct ForLoopBody {  // structs/function.rs:268-274
  bool is_anon;
  std::string name;
  IterExprList outputs, inputs;
  uint64_t instance_count = 0, witness_count = 0;
  std::vector<Gate> subcircuit;
};

struct Gate {  // structs/gates.rs:17-55; tag = DirectiveSet value
  enum Tag {
    Constant = 1, AssertZero, Copy, Add, Mul, AddConstant, MulConstant, And, Xor, Not,
    Instance, Witness, Free, Call, AnonCall, Switch, For
  } tag;
  WireId out = 0, left = 0, right = 0;  // generic wire fields (inp -> left)
  Value constant;
  bool has_last = false;                 // Free(first=left, last=right?)
  std::string name;                      // Call name / For iterator
  WireList output_wires, input_wires;
  uint64_t instance_count = 0, witness_count = 0;
  std::vector<Gate> subcircuit;
  std::vector<Value> cases;
  std::vector<CaseInvoke> branches;
  uint64_t for_first = 0, for_last = 0;
  std::shared_ptr<ForLoopBody> body;
};

struct Header {  // structs/header.rs:11-15
  std::string version;
  Value field_characteristic;
  uint32_t field_degree = 0;
};
struct Function {  // structs/function.rs:17-25
  std::string name;
  uint64_t output_count = 0, input_count = 0, instance_count = 0, witness_count = 0;
  std::vector<Gate> body;
};
struct Relation {  // structs/relation.rs:34-41
  Header header;
  uint16_t gate_mask = 0, feat_mask = 0;
  std::vector<Function> functions;
  std::vector<Gate> gates;
};
struct Instance { Header header; std::vector<Value> common_inputs; };
struct Witness { Header header; std::vector<Value> short_witness; };
struct Message {  // structs/message.rs:8-13
  enum Kind { IsRelation = 1, IsInstance = 2, IsWitness = 3 } kind;
  Relation relation;
  Instance instance;
  Witness witness;
};

// masks: structs/relation.rs:15-32
enum : uint16_t {
  M_ADD = 0x0001, M_ADDC = 0x0002, M_MUL = 0x0004, M_MULC = 0x0008, M_ARITH = 0x000F,
  M_XOR = 0x0100, M_AND = 0x0200, M_NOT = 0x0400, M_BOOL = 0x0700,
  M_FUNCTION = 0x1000, M_FOR = 0x2000, M_SWITCH = 0x4000, M_SIMPLE = 0
};

static inline std::string strip_spaces(const std::string& s) {
  std::string r;
  for (char c : s) if (c != ' ') r.push_back(c);
  return r;
}
static inline std::vector<std::string> split_commas(const std::string& s) {
  std::vector<std::string> v;
  size_t b = 0;
  for (;;) {
    size_t e = s.find(',', b);
    v.push_back(s.substr(b, e == std::string::npos ? std::string::npos : e - b));
    if (e == std::string::npos) break;
    b = e + 1;
  }
  return v;
}
// structs/relation.rs:144-167
static inline uint16_t parse_gate_set(const std::string& gateset) {
  uint16_t ret = 0;
  for (const std::string& raw : split_commas(gateset)) {
    const std::string s = strip_spaces(raw);
    if (s == "arithmetic") return M_ARITH;
    else if (s == "@add") ret |= M_ADD;
    else if (s == "@addc") ret |= M_ADDC;
    else if (s == "@mul") ret |= M_MUL;
    else if (s == "@mulc") ret |= M_MULC;
    else if (s == "boolean") return M_BOOL;
    else if (s == "@xor") ret |= M_XOR;
    else if (s == "@not") ret |= M_NOT;
    else if (s == "@and") ret |= M_AND;
    else if (s == "") {}
    else throw Err("Unable to parse the following gateset: " + gateset);
  }
  return ret;
}
// structs/relation.rs:229-244
static inline uint16_t parse_feature_toggle(const std::string& features) {
  uint16_t ret = 0;
  for (const std::string& raw : split_commas(features)) {
    const std::string s = strip_spaces(raw);
    if (s == "@function") ret |= M_FUNCTION;
    else if (s == "@for") ret |= M_FOR;
    else if (s == "@switch") ret |= M_SWITCH;
    else if (s == "simple") return M_SIMPLE;
    else if (s == "") {}
    else throw Err("Unable to parse following feature toggles " + raw);
  }
  return ret;
}

// ---------------- FlatBuffers walking ----------------
struct Buf {
  const uint8_t* p;
  size_t n;
  void need(size_t off, size_t len) const {
    if (off > n || len > n - off) throw Err("panic: flatbuffer access out of bounds");
  }
  uint8_t u8(size_t o) const { need(o, 1); return p[o]; }
  uint16_t u16(size_t o) const { need(o, 2); uint16_t v; memcpy(&v, p + o, 2); return v; }
  uint32_t u32(size_t o) const { need(o, 4); uint32_t v; memcpy(&v, p + o, 4); return v; }
  int32_t i32(size_t o) const { need(o, 4); int32_t v; memcpy(&v, p + o, 4); return v; }
  uint64_t u64(size_t o) const { need(o, 8); uint64_t v; memcpy(&v, p + o, 8); return v; }
};

struct Table {
  const Buf* b = nullptr;
  size_t pos = 0;
  bool ok() const { return b != nullptr; }
  // byte offset of field `slot` (a VT_* constant) inside the table, 0 if absent
  size_t field(unsigned slot) const {
    if (!b) throw Err("panic: called `Option::unwrap()` on a `None` value");
    const size_t vt = (size_t)((int64_t)pos - b->i32(pos));
    const unsigned vtsize = b->u16(vt);
    if (slot + 2 > vtsize) return 0;
    const unsigned off = b->u16(vt + slot);
    return off ? pos + off : 0;
  }
  uint8_t get_u8(unsigned slot) const { size_t f = field(slot); return f ? b->u8(f) : 0; }
  uint32_t get_u32(unsigned slot) const { size_t f = field(slot); return f ? b->u32(f) : 0; }
  uint64_t get_u64(unsigned slot) const { size_t f = field(slot); return f ? b->u64(f) : 0; }
  Table get_table(unsigned slot) const {
    size_t f = field(slot);
    Table t;
    if (f) { t.b = b; t.pos = f + b->u32(f); }
    return t;
  }
  // vectors / strings: returns position of the length word, 0 if absent
  size_t get_vec(unsigned slot) const { size_t f = field(slot); return f ? f + b->u32(f) : 0; }
  bool get_string(unsigned slot, std::string& out) const {
    size_t v = get_vec(slot);
    if (!v) return false;
    uint32_t len = b->u32(v);
    b->need(v + 4, len);
    out.assign((const char*)b->p + v + 4, len);
    return true;
  }
  bool get_bytes(unsigned slot, Value& out) const {
    size_t v = get_vec(slot);
    if (!v) return false;
    uint32_t len = b->u32(v);
    b->need(v + 4, len);
    out.assign(b->p + v + 4, b->p + v + 4 + len);
    return true;
  }
};
static inline uint32_t vec_len(const Buf& b, size_t v) {
  const uint32_t n = b.u32(v);
  if ((size_t)n > b.n - (v + 4 <= b.n ? v + 4 : b.n)) throw Err("panic: flatbuffer vector length exceeds the message");
  return n;
}
static inline Table vec_table(const Buf& b, size_t v, uint32_t i) {
  size_t e = v + 4 + 4 * (size_t)i;
  Table t;
  t.b = &b;
  t.pos = e + b.u32(e);
  return t;
}

static inline WireId wire_id(const Table& t) { return t.get_u64(4); }
static inline WireId req_wire(const Table& gate, unsigned slot, const char* missing) {
  Table w = gate.get_table(slot);
  if (!w.ok()) throw Err(missing);
  return wire_id(w);
}
static inline Value value_from(const Table& v) {  // structs/value.rs:14-16
  Value out;
  if (!v.get_bytes(4, out)) throw Err("Missing value");
  return out;
}
static inline std::vector<Value> values_vector(const Buf& b, size_t v) {
  std::vector<Value> out;
  for (uint32_t i = 0; i < vec_len(b, v); ++i) out.push_back(value_from(vec_table(b, v, i)));
  return out;
}

// structs/wire.rs:90-157
static inline WireList wirelist_from(const Table& t) {
  WireList out;
  size_t v = t.get_vec(4);
  if (!v) throw Err("Missing wire list");
  for (uint32_t i = 0; i < vec_len(*t.b, v); ++i) {
    Table el = vec_table(*t.b, v, i);
    uint8_t ty = el.get_u8(4);
    Table inner = el.get_table(6);
    WireListElement e;
    if (ty == 1) {
      e.is_range = false;
      e.first = e.last = wire_id(inner);
    } else if (ty == 2) {
      e.is_range = true;
      Table f = inner.get_table(4), l = inner.get_table(6);
      if (!f.ok()) throw Err("Missing first value of range");
      if (!l.ok()) throw Err("Missing last value of range");
      e.first = wire_id(f);
      e.last = wire_id(l);
    } else {
      throw Err("Unknown type in WireListElement");
    }
    out.push_back(e);
  }
  return out;
}

// structs/iterators.rs:35-114
// guards of the test machine (a corrupt message may hold a zero / cyclic offset)
struct OracleNesting {
  static int& depth() { static thread_local int d = 0; return d; }
  OracleNesting() { if (++depth() > 256) { --depth(); throw Err("panic: tables nested deeper than 256 levels"); } }
  ~OracleNesting() { --depth(); }
};

static inline IterExprWireNumber iterexpr_from(const Table& t) {
  OracleNesting guard;
  IterExprWireNumber r;
  uint8_t ty = t.get_u8(4);
  Table v = t.get_table(6);
  auto sub = [&](unsigned slot, const char* missing) {
    Table s = v.get_table(slot);
    if (!s.ok()) throw Err(missing);
    return std::make_shared<IterExprWireNumber>(iterexpr_from(s));
  };
  switch (ty) {
    case 1: r.kind = IterExprWireNumber::Const; r.value = v.get_u64(4); break;
    case 2:
      r.kind = IterExprWireNumber::Name;
      if (!v.get_string(4, r.name)) throw Err("IterExpr: No name given");
      break;
    case 3: r.kind = IterExprWireNumber::Add; r.left = sub(4, "Missing left operand"); r.right = sub(6, "Missing right operand"); break;
    case 4: r.kind = IterExprWireNumber::Sub; r.left = sub(4, "Missing left operand"); r.right = sub(6, "Missing right operand"); break;
    case 5: r.kind = IterExprWireNumber::Mul; r.left = sub(4, "Missing left operand"); r.right = sub(6, "Missing right operand"); break;
    case 6: r.kind = IterExprWireNumber::DivConst; r.left = sub(4, "Missing numerator"); r.value = v.get_u64(6); break;
    default: throw Err("Unknown Iterator Expression type");
  }
  return r;
}
// structs/iterators.rs:244-270,314-326
static inline IterExprList iterexprlist_from(const Table& t) {
  IterExprList out;
  size_t v = t.get_vec(4);
  if (!v) throw Err("Missing wire list");
  for (uint32_t i = 0; i < vec_len(*t.b, v); ++i) {
    Table el = vec_table(*t.b, v, i);
    uint8_t ty = el.get_u8(4);
    Table inner = el.get_table(6);
    IterExprListElement e;
    if (ty == 1) {
      e.is_range = false;
      e.first = iterexpr_from(inner);
    } else if (ty == 2) {
      e.is_range = true;
      Table f = inner.get_table(4), l = inner.get_table(6);
      if (!f.ok()) throw Err("Missing first value of range");
      if (!l.ok()) throw Err("Missing last value of range");
      e.first = iterexpr_from(f);
      e.last = iterexpr_from(l);
    } else {
      throw Err("Unknown type in IterExprWireListElement");
    }
    out.push_back(e);
  }
  return out;
}

static inline std::vector<Gate> gates_vector(const Buf& b, size_t v);

// structs/gates.rs:60-259
static inline Gate gate_from(const Table& d) {
  OracleNesting guard;
  Gate g;
  const uint8_t ty = d.get_u8(4);
  const Table t = d.get_table(6);
  if (ty == 0) throw Err("No gate type");
  g.tag = (Gate::Tag)ty;
  auto req_list = [&](const Table& tb, unsigned slot, const char* missing) {
    Table l = tb.get_table(slot);
    if (!l.ok()) throw Err(missing);
    return wirelist_from(l);
  };
  switch (ty) {
    case Gate::Constant:
      g.out = req_wire(t, 4, "Missing output");
      if (!t.get_bytes(6, g.constant)) throw Err("Missing constant");
      break;
    case Gate::AssertZero: g.left = req_wire(t, 4, "Missing input"); break;
    case Gate::Copy:
    case Gate::Not:
      g.out = req_wire(t, 4, "Missing output");
      g.left = req_wire(t, 6, "Missing input");
      break;
    case Gate::Add:
    case Gate::Mul:
    case Gate::And:
    case Gate::Xor:
      g.out = req_wire(t, 4, "Missing output");
      g.left = req_wire(t, 6, "Missing left input");
      g.right = req_wire(t, 8, "Missing right input");
      break;
    case Gate::AddConstant:
    case Gate::MulConstant:
      g.out = req_wire(t, 4, "Missing output");
      g.left = req_wire(t, 6, "Missing input");
      if (!t.get_bytes(8, g.constant)) throw Err("Missing constant");
      break;
    case Gate::Instance:
    case Gate::Witness: g.out = req_wire(t, 4, "Missing output"); break;
    case Gate::Free: {
      g.left = req_wire(t, 4, "Missing first wire");
      Table l = t.get_table(6);
      g.has_last = l.ok();
      if (g.has_last) g.right = wire_id(l);
      break;
    }
    case Gate::Call:
      if (!t.get_string(4, g.name)) throw Err("Missing function name.");
      g.output_wires = req_list(t, 6, "Missing outputs");
      g.input_wires = req_list(t, 8, "Missing inputs");
      break;
    case Gate::AnonCall: {
      Table inner = t.get_table(6);
      if (!inner.ok()) throw Err("Missing inner AbstractAnonCall");
      g.output_wires = req_list(t, 4, "Missing output wires");
      g.input_wires = req_list(inner, 4, "Missing input wires");
      g.instance_count = inner.get_u64(6);
      g.witness_count = inner.get_u64(8);
      size_t sv = inner.get_vec(10);
      if (!sv) throw Err("Missing subcircuit");
      g.subcircuit = gates_vector(*t.b, sv);
      break;
    }
    case Gate::Switch: {
      size_t cv = t.get_vec(8);
      if (!cv) throw Err("Missing cases values");
      g.cases = values_vector(*t.b, cv);
      g.left = req_wire(t, 4, "Missing condition wire.");
      g.output_wires = req_list(t, 6, "Missing output wires");
      size_t bv = t.get_vec(10);
      if (!bv) throw Err("Missing branches");
      for (uint32_t i = 0; i < vec_len(*t.b, bv); ++i) {  // structs/function.rs:132-172
        Table ci = vec_table(*t.b, bv, i);
        uint8_t ity = ci.get_u8(4);
        Table inv = ci.get_table(6);
        CaseInvoke c;
        if (ity == 1) {
          c.is_anon = false;
          if (!inv.get_string(4, c.name)) throw Err("Missing function name.");
          c.input_wires = req_list(inv, 6, "Missing inputs");
        } else if (ity == 2) {
          c.is_anon = true;
          size_t sv = inv.get_vec(10);
          if (!sv) throw Err("Missing implementation");
          c.subcircuit = gates_vector(*t.b, sv);
          c.input_wires = req_list(inv, 4, "Missing inputs");
          c.instance_count = inv.get_u64(6);
          c.witness_count = inv.get_u64(8);
        } else {
          throw Err("No directive type");
        }
        g.branches.push_back(std::move(c));
      }
      break;
    }
    case Gate::For: {
      g.output_wires = req_list(t, 4, "missing output list");
      const uint8_t bty = t.get_u8(12);
      Table body = t.get_table(14);
      auto fb = std::make_shared<ForLoopBody>();
      auto req_iter = [&](unsigned slot, const char* missing) {
        Table l = body.get_table(slot);
        if (!l.ok()) throw Err(missing);
        return iterexprlist_from(l);
      };
      if (bty == 1) {
        fb->is_anon = false;
        if (!body.get_string(4, fb->name)) throw Err("Missing function in function name");
        fb->outputs = req_iter(6, "missing output list");
        fb->inputs = req_iter(8, "missing input list");
      } else if (bty == 2) {
        fb->is_anon = true;
        fb->outputs = req_iter(4, "missing output list");
        fb->inputs = req_iter(6, "missing input list");
        fb->instance_count = body.get_u64(8);
        fb->witness_count = body.get_u64(10);
        size_t sv = body.get_vec(12);
        if (!sv) throw Err("Missing body");
        fb->subcircuit = gates_vector(*t.b, sv);
      } else {
        throw Err("Unknown body type");
      }
      g.body = fb;
      if (!t.get_string(6, g.name)) throw Err("Missing iterator name");
      g.for_first = t.get_u64(8);
      g.for_last = t.get_u64(10);
      break;
    }
    default: throw Err("No gate type");
  }
  return g;
}
static inline std::vector<Gate> gates_vector(const Buf& b, size_t v) {
  std::vector<Gate> out;
  const uint32_t n = vec_len(b, v);
  out.reserve(n);
  for (uint32_t i = 0; i < n; ++i) out.push_back(gate_from(vec_table(b, v, i)));
  return out;
}

static inline Header header_from(const Table& h) {  // structs/header.rs:37-56
  if (!h.ok()) throw Err("Missing header");
  Header r;
  if (!h.get_string(4, r.version)) throw Err("Missing version");
  Table fc = h.get_table(6);
  if (!fc.ok()) throw Err("Missing field characteristic");
  r.field_characteristic = value_from(fc);
  r.field_degree = h.get_u32(8);
  return r;
}

// One size-prefixed message (structs/message.rs:15-37).  `p` points at the
// 4-byte size prefix; `n` = prefix + body length.
static inline Message message_from(const uint8_t* p, size_t n) {
  Buf b{p + 4, n - 4};
  Table root;
  root.b = &b;
  root.pos = b.u32(0);
  const uint8_t ty = root.get_u8(4);
  Table m = root.get_table(6);
  Message msg;
  if (ty == 1) {
    msg.kind = Message::IsRelation;
    if (!m.ok()) throw Err("Invalid message.");
    Relation& r = msg.relation;  // structs/relation.rs:47-72
    size_t gv = m.get_vec(12);
    if (!gv) throw Err("Missing directives");
    size_t fv = m.get_vec(10);
    if (fv)
      for (uint32_t i = 0; i < vec_len(b, fv); ++i) {
        Table ft = vec_table(b, fv, i);
        Function f;
        size_t bv = ft.get_vec(14);
        if (!bv) throw Err("Missing reference implementation");
        if (!ft.get_string(4, f.name)) throw Err("Missing name");
        f.output_count = ft.get_u64(6);
        f.input_count = ft.get_u64(8);
        f.instance_count = ft.get_u64(10);
        f.witness_count = ft.get_u64(12);
        f.body = gates_vector(b, bv);
        r.functions.push_back(std::move(f));
      }
    r.header = header_from(m.get_table(4));
    std::string s;
    if (!m.get_string(6, s)) throw Err("Missing gateset description");
    r.gate_mask = parse_gate_set(s);
    if (!m.get_string(8, s)) throw Err("Missing feature toggles");
    r.feat_mask = parse_feature_toggle(s);
    r.gates = gates_vector(b, gv);
  } else if (ty == 2) {
    msg.kind = Message::IsInstance;
    if (!m.ok()) throw Err("Invalid message.");
    msg.instance.header = header_from(m.get_table(4));
    size_t v = m.get_vec(6);
    if (!v) throw Err("Missing common_input");
    msg.instance.common_inputs = values_vector(b, v);
  } else if (ty == 3) {
    msg.kind = Message::IsWitness;
    if (!m.ok()) throw Err("Invalid message.");
    msg.witness.header = header_from(m.get_table(4));
    size_t v = m.get_vec(6);
    if (!v) throw Err("Missing short_witness");
    msg.witness.short_witness = values_vector(b, v);
  } else {
    throw Err("Invalid message type");
  }
  return msg;
}

// consumers/utils.rs:6-25 split_messages / read_buffer framing
static inline std::vector<std::pair<size_t, size_t>> split_messages(const uint8_t* p, size_t n) {
  std::vector<std::pair<size_t, size_t>> out;
  size_t off = 0;
  for (;;) {
    if (n - off < 4) break;
    uint32_t sz;
    memcpy(&sz, p + off, 4);
    size_t total = 4 + (size_t)sz;
    if (total <= 4) break;
    if (total > n - off) break;  // truncated stream: read_exact fails -> end
    out.push_back({off, total});
    off += total;
  }
  return out;
}

}  // namespace zko
