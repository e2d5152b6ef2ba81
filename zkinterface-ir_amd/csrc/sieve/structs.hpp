// Owned SIEVE IR structures of the product host: the C++ mirror of the
// reference's rust/src/structs/ (Message, Messages, Header, Instance, Witness,
// Relation, Gate, Function, CaseInvoke, ForLoopBody, WireList, IterExpr*;
// re-exported at rust/src/lib.rs:41-44).  Field names and meaning follow the
// reference so that the Evaluator reads like consumers/evaluator.rs.
#pragma once
#include <stdint.h>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace zki {

using WireId = uint64_t;            // rust/src/structs/mod.rs: WireId = u64
using Value = std::vector<uint8_t>; // little-endian bytes, any length (sieve_ir.fbs:55-59)

// Error carrying the reference's message text (lib.rs:47 `Result<T, Box<dyn Error>>`).
struct Error : std::runtime_error {
  explicit Error(const std::string& s) : std::runtime_error(s) {}
};
// Conditions on which the reference panics (unwrap/panic!) rather than returning Err.
struct Panic : std::runtime_error {
  explicit Panic(const std::string& s) : std::runtime_error(s) {}
};

// structs/relation.rs:15-32
namespace mask {
constexpr uint16_t ADD = 0x0001, ADDC = 0x0002, MUL = 0x0004, MULC = 0x0008, ARITH = ADD | ADDC | MUL | MULC;
constexpr uint16_t XOR = 0x0100, AND = 0x0200, NOT = 0x0400, BOOL = XOR | AND | NOT;
constexpr uint16_t FUNCTION = 0x1000, FOR = 0x2000, SWITCH = 0x4000, SIMPLE = 0x0000;
inline bool contains_feature(uint16_t set, uint16_t feature) { return (set & feature) == feature; }  // :284-286
}  // namespace mask

struct WireRange {  // a WireListElement: single wire (first == last, !range) or inclusive range
  WireId first = 0, last = 0;
  bool range = false;
};
using WireList = std::vector<WireRange>;

struct IterExpr {  // structs/iterators.rs:17-30
  enum Op : uint8_t { CONST = 1, NAME = 2, ADD = 3, SUB = 4, MUL = 5, DIV_CONST = 6 };
  Op op = CONST;
  uint64_t value = 0;  // CONST literal, DIV_CONST denominator
  std::string name;    // NAME
  std::vector<IterExpr> args;  // operands (1 for DIV_CONST, 2 for ADD/SUB/MUL)
};
struct IterExprRange {  // IterExprListElement: Single(first) or Range(first, last)
  IterExpr first, last;
  bool range = false;
};
using IterExprList = std::vector<IterExprRange>;

struct Gate;
using Subcircuit = std::vector<Gate>;

struct CaseInvoke {  // structs/function.rs:120-131
  bool anonymous = false;
  std::string name;           // AbstractGateCall
  WireList input_wires;
  uint64_t instance_count = 0, witness_count = 0;  // AbstractAnonCall
  std::shared_ptr<Subcircuit> subcircuit;
};

struct ForLoopBody {  // structs/function.rs:268-274
  bool anonymous = false;
  std::string name;           // IterExprCall
  IterExprList outputs, inputs;
  uint64_t instance_count = 0, witness_count = 0;  // IterExprAnonCall
  std::shared_ptr<Subcircuit> subcircuit;
};

enum class GateKind : uint8_t {  // == DirectiveSet tags (sieve_ir_generated.rs:422-440)
  None = 0, Constant, AssertZero, Copy, Add, Mul, AddConstant, MulConstant, And, Xor, Not,
  Instance, Witness, Free, Call, AnonCall, Switch, For
};

// Everything only the structured gates need lives behind one pointer so that a
// simple gate stays 32 bytes (a 1M-gate relation is ~32 MB, not ~200 MB).
struct GateExt {
  Value constant;                     // Constant / AddConstant / MulConstant
  std::string name;                   // Call: function; For: iterator
  WireList output_wires, input_wires; // Call / AnonCall / Switch / For(global outputs)
  uint64_t instance_count = 0, witness_count = 0;
  std::shared_ptr<Subcircuit> subcircuit;  // AnonCall
  std::vector<Value> cases;           // Switch
  std::vector<CaseInvoke> branches;   // Switch
  uint64_t first = 0, last = 0;       // For bounds (inclusive)
  ForLoopBody body;                   // For
};

struct Gate {  // structs/gates.rs:17-55
  GateKind kind = GateKind::None;
  bool has_last = false;  // Free(first, Some(last))
  WireId out = 0;         // output (or Free.first, AssertZero.input, Switch.condition in `in0`)
  WireId in0 = 0, in1 = 0;
  std::shared_ptr<GateExt> ext;
};

struct Header {  // structs/header.rs:11-15
  std::string version;
  Value field_characteristic;
  uint32_t field_degree = 0;
};
struct Function {  // structs/function.rs:17-25
  std::string name;
  uint64_t output_count = 0, input_count = 0, instance_count = 0, witness_count = 0;
  std::shared_ptr<Subcircuit> body;
};
struct Relation {  // structs/relation.rs:34-41
  Header header;
  uint16_t gate_mask = 0, feat_mask = 0;
  std::vector<Function> functions;
  std::vector<Gate> gates;
};
struct Instance {  // structs/instance.rs:12-16
  Header header;
  std::vector<Value> common_inputs;
};
struct Witness {  // structs/witness.rs:12-16
  Header header;
  std::vector<Value> short_witness;
};
struct Message {  // structs/message.rs:8-13
  enum Kind : uint8_t { None = 0, IsRelation = 1, IsInstance = 2, IsWitness = 3 };
  Kind kind = None;
  Instance instance;
  Witness witness;
  Relation relation;
};
struct Messages {  // structs/messages.rs:4-19
  std::vector<Instance> instances;
  std::vector<Witness> witnesses;
  std::vector<Relation> relations;
  void push_message(Message&& m) {
    switch (m.kind) {
      case Message::IsInstance: instances.push_back(std::move(m.instance)); break;
      case Message::IsWitness: witnesses.push_back(std::move(m.witness)); break;
      case Message::IsRelation: relations.push_back(std::move(m.relation)); break;
      default: break;
    }
  }
};

}  // namespace zki
