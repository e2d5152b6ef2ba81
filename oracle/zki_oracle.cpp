// TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: a literal CPU
// restatement of the reference's `Evaluator` + `PlaintextBackend`
// (/root/reference/rust/src/consumers/evaluator.rs).  Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
// product (zkinterface-ir_amd/) never does.
//
// Pinned by the reference's own fixtures and known answers (tests/test_oracle_*):
//   rust/examples/00{0,1,2}_*.sieve, evaluator.rs:950-984 (modexp KATs),
//   evaluator.rs:987-1004,1083-1104, boolean_examples.rs, builder.rs tests,
//   and the backend-op trace digests of SURVEY.md Appendix A.
//
// Data structures are kept as in the reference on purpose (hash-map scope per
// (sub)circuit keyed by u64, heap big integers, `%` by long division on every
// gate, deep-copied function bodies), because this file is also the "port"
// CPU baseline timed by bench.py.
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <deque>
#include <fstream>
#include <functional>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "biguint.hpp"
#include "sieve_oracle_reader.hpp"

namespace zko {

// ---- trace of value-returning ZKBackend calls ----
enum TraceKind : uint8_t {
  T_COPY = 0, T_CONSTANT, T_ADD, T_MUL, T_ADDC, T_MULC, T_AND, T_XOR, T_NOT, T_INSTANCE, T_WITNESS
};
static const char* kTraceNames[] = {"copy", "constant", "add", "mul", "addc", "mulc",
                                    "and", "xor", "not", "instance", "witness"};

// evaluator.rs:848-947
struct PlaintextBackend {
  BigUint m;
  bool trace_on = false;
  std::vector<uint8_t> trace_kind;
  std::vector<BigUint> trace_val;
  uint64_t n_ops = 0, n_asserts = 0;

  uint64_t max_ops = 1ull << 30;  // guard for the test machine only: a corrupt loop bound unrolls forever
  BigUint rec(TraceKind k, BigUint v) {
    if (++n_ops > max_ops) throw Err("oracle guard: more than max_ops backend operations");
    if (trace_on) {
      trace_kind.push_back(k);
      trace_val.push_back(v);
    }
    return v;
  }
  static BigUint from_bytes_le(const Value& v) { return BigUint::from_bytes_le(v); }  // :862-864
  void set_field(const Value& modulus, uint32_t degree, bool) {                       // :866-875
    m = BigUint::from_bytes_le(modulus);
    if (m.is_zero()) throw Err("Modulus cannot be zero.");
    if (degree != 1) throw Err("Field should be of degree 1");
  }
  BigUint one() const { return BigUint(1); }
  BigUint minus_one() const {  // :881-886
    if (m.is_zero()) throw Err("Modulus is not initiated, used `set_field()` before calling.");
    return BigUint::sub(m, one());
  }
  BigUint zero() const { return BigUint(); }
  BigUint copy(const BigUint& w) { return rec(T_COPY, w); }                 // :892-894
  BigUint constant(const BigUint& v) { return rec(T_CONSTANT, v); }         // :896-898
  void assert_zero(const BigUint& w) {                                      // :900-906
    ++n_asserts;
    if (!w.is_zero()) throw Err("AssertZero failed");
  }
  BigUint add(const BigUint& a, const BigUint& b) { return rec(T_ADD, BigUint::rem(BigUint::add(a, b), m)); }
  BigUint multiply(const BigUint& a, const BigUint& b) { return rec(T_MUL, BigUint::rem(BigUint::mul(a, b), m)); }
  BigUint add_constant(const BigUint& a, const BigUint& b) { return rec(T_ADDC, BigUint::rem(BigUint::add(a, b), m)); }
  BigUint mul_constant(const BigUint& a, const BigUint& b) { return rec(T_MULC, BigUint::rem(BigUint::mul(a, b), m)); }
  BigUint and_(const BigUint& a, const BigUint& b) { return rec(T_AND, BigUint::rem(BigUint::bitand_(a, b), m)); }
  BigUint xor_(const BigUint& a, const BigUint& b) { return rec(T_XOR, BigUint::rem(BigUint::bitxor_(a, b), m)); }
  BigUint not_(const BigUint& a) { return rec(T_NOT, a.is_zero() ? BigUint(1) : BigUint()); }  // :932-938
  BigUint instance(const BigUint& v) { return rec(T_INSTANCE, v); }         // :940-942 (via constant)
  BigUint witness(const BigUint* v) {                                       // :944-946
    if (!v) throw Err("panic: Missing witness value for PlaintextBackend");
    return rec(T_WITNESS, *v);
  }
};

typedef std::unordered_map<WireId, BigUint> Scope;
typedef std::deque<BigUint> Queue;
typedef std::unordered_map<std::string, uint64_t> Iterators;

struct FunctionDeclaration {  // evaluator.rs:130-136
  std::vector<Gate> subcircuit;
  size_t instance_nbr, witness_nbr, output_count, input_count;
};
typedef std::unordered_map<std::string, FunctionDeclaration> Functions;

// evaluator.rs:775-797
static void set(Scope& scope, WireId id, BigUint w) {
  auto it = scope.find(id);
  if (it != scope.end()) {
    it->second = std::move(w);  // HashMap::insert replaces, then reports
    throw Err("Wire_" + std::to_string(id) + " already has a value in this scope.");
  }
  scope.emplace(id, std::move(w));
}
static const BigUint& get(const Scope& scope, WireId id) {
  auto it = scope.find(id);
  if (it == scope.end()) throw Err("No value given for wire_" + std::to_string(id));
  return it->second;
}
static void remove(Scope& scope, WireId id) {
  if (scope.erase(id) == 0) throw Err("No value given for wire_" + std::to_string(id));
}

// structs/wire.rs:178-203
static std::vector<WireId> expand_wirelist(const WireList& wl) {
  std::vector<WireId> out;
  for (const WireListElement& e : wl) {
    if (!e.is_range) {
      out.push_back(e.first);
    } else {
      if (e.last <= e.first)
        throw Err("In WireRange, last WireId (" + std::to_string(e.last) +
                  ") must be strictly greater than first WireId (" + std::to_string(e.first) + ").");
      if (e.last - e.first >= (1ull << 28)) throw Err("oracle guard: wire list expands to more than 2^28 wires");
      for (WireId w = e.first;; ++w) {
        out.push_back(w);
        if (w == e.last) break;
      }
    }
  }
  return out;
}
// structs/iterators.rs:349-403 (u64 arithmetic wraps as in a release build)
static uint64_t evaluate_iterexpr(const IterExprWireNumber& e, const Iterators& known) {
  switch (e.kind) {
    case IterExprWireNumber::Const: return e.value;
    case IterExprWireNumber::Name: {
      auto it = known.find(e.name);
      if (it == known.end()) throw Err("panic: Unknown iterator name " + e.name);
      return it->second;
    }
    case IterExprWireNumber::Add: return evaluate_iterexpr(*e.left, known) + evaluate_iterexpr(*e.right, known);
    case IterExprWireNumber::Sub: return evaluate_iterexpr(*e.left, known) - evaluate_iterexpr(*e.right, known);
    case IterExprWireNumber::Mul: return evaluate_iterexpr(*e.left, known) * evaluate_iterexpr(*e.right, known);
    case IterExprWireNumber::DivConst: {
      uint64_t n = evaluate_iterexpr(*e.left, known);
      if (e.value == 0) throw Err("panic: attempt to divide by zero");
      return n / e.value;
    }
  }
  throw Err("panic: bad iterexpr");
}
static std::vector<WireId> evaluate_iterexpr_list(const IterExprList& l, const Iterators& known) {
  std::vector<WireId> out;
  for (const IterExprListElement& e : l) {
    if (!e.is_range) {
      out.push_back(evaluate_iterexpr(e.first, known));
    } else {
      uint64_t a = evaluate_iterexpr(e.first, known), b = evaluate_iterexpr(e.last, known);
      if (a <= b && b - a >= (1ull << 28)) throw Err("oracle guard: wire list expands to more than 2^28 wires");
      if (a <= b)
        for (uint64_t w = a;; ++w) {
          out.push_back(w);
          if (w == b) break;
        }
    }
  }
  return out;
}

struct Ctx {
  PlaintextBackend* backend;
  const Functions* known_functions;
  const BigUint* modulus;
  bool is_boolean;
};

// evaluator.rs:80-126
static BigUint as_mul(Ctx& c, const BigUint& a, const BigUint& b) { return c.is_boolean ? c.backend->and_(a, b) : c.backend->multiply(a, b); }
static BigUint as_add(Ctx& c, const BigUint& a, const BigUint& b) { return c.is_boolean ? c.backend->xor_(a, b) : c.backend->add(a, b); }
static BigUint as_negate(Ctx& c, const BigUint& w) { return c.is_boolean ? c.backend->copy(w) : c.backend->mul_constant(w, c.backend->minus_one()); }
static BigUint as_add_one(Ctx& c, const BigUint& w) { return c.is_boolean ? c.backend->not_(w) : c.backend->add_constant(w, c.backend->one()); }

// evaluator.rs:801-820
static BigUint exp_(Ctx& c, const BigUint& base, const BigUint& exponent) {
  if (exponent.is_one()) return c.backend->copy(base);
  BigUint previous = exp_(c, base, exponent.shr1());
  BigUint ret = as_mul(c, previous, previous);
  if (exponent.is_odd()) return as_mul(c, ret, base);
  return ret;
}
// evaluator.rs:823-839
static BigUint compute_weight(Ctx& c, const Value& case_, const BigUint& condition) {
  BigUint case_wire = c.backend->constant(PlaintextBackend::from_bytes_le(case_));
  BigUint exponent = BigUint::sub(*c.modulus, BigUint(1));
  BigUint minus_cond = as_negate(c, condition);
  BigUint base = as_add(c, case_wire, minus_cond);
  BigUint base_to_exp = exp_(c, base, exponent);
  BigUint right = as_negate(c, base_to_exp);
  return as_add_one(c, right);
}

static void ingest_gate(const Gate& gate, Ctx& c, Scope& scope, Iterators& known_iterators, Queue& instances,
                        Queue& witnesses, const BigUint* weight);

// evaluator.rs:698-746
static void ingest_subcircuit(const std::vector<Gate>& subcircuit, Ctx& c, const std::vector<WireId>& output_list,
                              const std::vector<WireId>& input_list, Scope& scope, Iterators& known_iterators,
                              Queue& instances, Queue& witnesses, const BigUint* weight) {
  struct Depth {  // guard of the test machine: self-recursive functions overflow the stack in the reference
    Depth() { if (++d() > 2000) { --d(); throw Err("subcircuits nested deeper than 2000 calls"); } }
    ~Depth() { --d(); }
    static int& d() { static thread_local int v = 0; return v; }
  } depth_guard;
  Scope new_scope;
  for (size_t idx = 0; idx < input_list.size(); ++idx) {
    const BigUint& i = get(scope, input_list[idx]);
    set(new_scope, (uint64_t)(idx + output_list.size()), c.backend->copy(i));
  }
  for (const Gate& g : subcircuit) ingest_gate(g, c, new_scope, known_iterators, instances, witnesses, weight);
  for (size_t idx = 0; idx < output_list.size(); ++idx) {
    const BigUint& w = get(new_scope, (uint64_t)idx);
    set(scope, output_list[idx], c.backend->copy(w));
  }
}

static std::string wrong_count(const char* what, const std::string& name, size_t expected, size_t got) {
  return std::string("Wrong number of ") + what + " variables in call to function " + name + " (Expected " +
         std::to_string(expected) + " / Got " + std::to_string(got) + ").";
}

// evaluator.rs:318-691
static void ingest_gate(const Gate& gate, Ctx& c, Scope& scope, Iterators& known_iterators, Queue& instances,
                        Queue& witnesses, const BigUint* weight) {
  PlaintextBackend& backend = *c.backend;
  switch (gate.tag) {
    case Gate::Constant: {
      BigUint wire = backend.constant(PlaintextBackend::from_bytes_le(gate.constant));
      set(scope, gate.out, std::move(wire));
      break;
    }
    case Gate::AssertZero: {
      const BigUint& inp_wire = get(scope, gate.left);
      BigUint should_be_zero = weight ? as_mul(c, *weight, inp_wire) : backend.copy(inp_wire);
      try {
        backend.assert_zero(should_be_zero);
      } catch (const Err&) {
        throw Err("Wire_" + std::to_string(gate.left) + " (may be weighted) should be 0, while it is not");
      }
      break;
    }
    case Gate::Copy: {
      BigUint out_wire = backend.copy(get(scope, gate.left));
      set(scope, gate.out, std::move(out_wire));
      break;
    }
    case Gate::Add: {
      const BigUint& l = get(scope, gate.left);
      const BigUint& r = get(scope, gate.right);
      set(scope, gate.out, backend.add(l, r));
      break;
    }
    case Gate::Mul: {
      const BigUint& l = get(scope, gate.left);
      const BigUint& r = get(scope, gate.right);
      set(scope, gate.out, backend.multiply(l, r));
      break;
    }
    case Gate::AddConstant: {
      const BigUint& l = get(scope, gate.left);
      BigUint r = PlaintextBackend::from_bytes_le(gate.constant);
      set(scope, gate.out, backend.add_constant(l, r));
      break;
    }
    case Gate::MulConstant: {
      const BigUint& l = get(scope, gate.left);
      BigUint r = PlaintextBackend::from_bytes_le(gate.constant);
      set(scope, gate.out, backend.mul_constant(l, r));
      break;
    }
    case Gate::And: {
      const BigUint& l = get(scope, gate.left);
      const BigUint& r = get(scope, gate.right);
      set(scope, gate.out, backend.and_(l, r));
      break;
    }
    case Gate::Xor: {
      const BigUint& l = get(scope, gate.left);
      const BigUint& r = get(scope, gate.right);
      set(scope, gate.out, backend.xor_(l, r));
      break;
    }
    case Gate::Not: {
      const BigUint& v = get(scope, gate.left);
      set(scope, gate.out, backend.not_(v));
      break;
    }
    case Gate::Instance: {
      if (instances.empty()) throw Err("Not enough instance to consume");
      BigUint val = std::move(instances.front());
      instances.pop_front();
      set(scope, gate.out, backend.instance(val));
      break;
    }
    case Gate::Witness: {
      if (witnesses.empty()) {
        set(scope, gate.out, backend.witness(nullptr));
      } else {
        BigUint val = std::move(witnesses.front());
        witnesses.pop_front();
        set(scope, gate.out, backend.witness(&val));
      }
      break;
    }
    case Gate::Free: {
      const WireId last_value = gate.has_last ? gate.right : gate.left;
      if (gate.left <= last_value)
        for (WireId cur = gate.left;; ++cur) {
          remove(scope, cur);
          if (cur == last_value) break;
        }
      break;
    }
    case Gate::Call: {
      auto it = c.known_functions->find(gate.name);
      if (it == c.known_functions->end()) throw Err("Unknown function");
      const FunctionDeclaration& f = it->second;
      std::vector<WireId> expanded_output = expand_wirelist(gate.output_wires);
      std::vector<WireId> expanded_input = expand_wirelist(gate.input_wires);
      if (expanded_output.size() != f.output_count)
        throw Err(wrong_count("output", gate.name, f.output_count, expanded_output.size()));
      if (expanded_input.size() != f.input_count)
        throw Err(wrong_count("input", gate.name, f.input_count, expanded_input.size()));
      Iterators fresh;
      ingest_subcircuit(f.subcircuit, c, expanded_output, expanded_input, scope, fresh, instances, witnesses, weight);
      break;
    }
    case Gate::AnonCall: {
      std::vector<WireId> expanded_output = expand_wirelist(gate.output_wires);
      std::vector<WireId> expanded_input = expand_wirelist(gate.input_wires);
      ingest_subcircuit(gate.subcircuit, c, expanded_output, expanded_input, scope, known_iterators, instances,
                        witnesses, weight);
      break;
    }
    case Gate::For: {
      if (gate.for_first <= gate.for_last)
        for (uint64_t i = gate.for_first;; ++i) {
          known_iterators[gate.name] = i;
          const ForLoopBody& body = *gate.body;
          if (!body.is_anon) {
            auto it = c.known_functions->find(body.name);
            if (it == c.known_functions->end()) throw Err("Unknown function");
            const FunctionDeclaration& f = it->second;
            std::vector<WireId> expanded_output = evaluate_iterexpr_list(body.outputs, known_iterators);
            std::vector<WireId> expanded_input = evaluate_iterexpr_list(body.inputs, known_iterators);
            if (expanded_output.size() != f.output_count)
              throw Err(wrong_count("output", body.name, f.output_count, expanded_output.size()));
            if (expanded_input.size() != f.input_count)
              throw Err(wrong_count("input", body.name, f.input_count, expanded_input.size()));
            Iterators fresh;
            ingest_subcircuit(f.subcircuit, c, expanded_output, expanded_input, scope, fresh, instances, witnesses,
                              weight);
          } else {
            std::vector<WireId> expanded_output = evaluate_iterexpr_list(body.outputs, known_iterators);
            std::vector<WireId> expanded_input = evaluate_iterexpr_list(body.inputs, known_iterators);
            ingest_subcircuit(body.subcircuit, c, expanded_output, expanded_input, scope, known_iterators, instances,
                              witnesses, weight);
          }
          if (i == gate.for_last) break;
        }
      known_iterators.erase(gate.name);
      break;
    }
    case Gate::Switch: {
      size_t max_instance_count = 0, max_witness_count = 0;
      for (const CaseInvoke& branch : gate.branches) {
        size_t ic, wc;
        if (!branch.is_anon) {
          auto it = c.known_functions->find(branch.name);
          if (it == c.known_functions->end()) throw Err("Unknown function");
          ic = it->second.instance_nbr;
          wc = it->second.witness_nbr;
        } else {
          ic = branch.instance_count;
          wc = branch.witness_count;
        }
        max_instance_count = std::max(max_instance_count, ic);
        max_witness_count = std::max(max_witness_count, wc);
      }
      // split_off + swap (evaluator.rs:586-591): the first min(len,max) values go to the branches
      Queue new_instances, new_witnesses;
      for (size_t k = std::min(instances.size(), max_instance_count); k > 0; --k) {
        new_instances.push_back(std::move(instances.front()));
        instances.pop_front();
      }
      for (size_t k = std::min(witnesses.size(), max_witness_count); k > 0; --k) {
        new_witnesses.push_back(std::move(witnesses.front()));
        witnesses.pop_front();
      }
      std::vector<Scope> branches_scope;
      std::vector<WireId> expanded_output = expand_wirelist(gate.output_wires);
      std::vector<BigUint> weights;
      const size_t n = std::min(gate.cases.size(), gate.branches.size());  // zip
      for (size_t k = 0; k < n; ++k) {
        const CaseInvoke& branch = gate.branches[k];
        BigUint branch_weight = compute_weight(c, gate.cases[k], get(scope, gate.left));
        BigUint weighted_branch_weight = weight ? as_mul(c, *weight, branch_weight) : branch_weight;
        Scope branch_scope;
        if (!branch.is_anon) {
          auto it = c.known_functions->find(branch.name);
          if (it == c.known_functions->end()) throw Err("Unknown function: " + branch.name);
          const FunctionDeclaration& f = it->second;
          std::vector<WireId> expanded_input = expand_wirelist(branch.input_wires);
          if (expanded_output.size() != f.output_count)
            throw Err(wrong_count("output", branch.name, f.output_count, expanded_output.size()));
          if (expanded_input.size() != f.input_count)
            throw Err(wrong_count("input", branch.name, f.input_count, expanded_input.size()));
          for (WireId wire : expanded_input) {
            const BigUint& w = get(scope, wire);
            branch_scope[wire] = backend.copy(w);  // HashMap::insert, silently replaces
          }
          Iterators fresh;
          Queue qi = new_instances, qw = new_witnesses;
          ingest_subcircuit(f.subcircuit, c, expanded_output, expanded_input, branch_scope, fresh, qi, qw,
                            &weighted_branch_weight);
        } else {
          std::vector<WireId> expanded_input = expand_wirelist(branch.input_wires);
          for (WireId wire : expanded_input) {
            const BigUint& w = get(scope, wire);
            branch_scope[wire] = backend.copy(w);
          }
          Queue qi = new_instances, qw = new_witnesses;
          ingest_subcircuit(branch.subcircuit, c, expanded_output, expanded_input, branch_scope, known_iterators, qi,
                            qw, &weighted_branch_weight);
        }
        weights.push_back(std::move(weighted_branch_weight));
        branches_scope.push_back(std::move(branch_scope));
      }
      for (WireId output_wire : expanded_output) {
        BigUint accu = backend.constant(backend.zero());
        for (size_t k = 0; k < branches_scope.size(); ++k) {
          BigUint weighted_wire = as_mul(c, get(branches_scope[k], output_wire), weights[k]);
          accu = as_add(c, accu, weighted_wire);
        }
        set(scope, output_wire, std::move(accu));
      }
      break;
    }
  }
}

// evaluator.rs:158-303
struct Evaluator {
  Scope values;
  BigUint modulus;
  Queue instance_queue, witness_queue;
  bool is_boolean = false;
  Functions known_functions;
  bool verified_at_least_one_gate = false;
  bool has_error = false;
  std::string found_error;
  bool panicked = false;

  void ingest_header(const Header& h) { modulus = BigUint::from_bytes_le(h.field_characteristic); }
  void ingest_instance(const Instance& i) {
    ingest_header(i.header);
    for (const Value& v : i.common_inputs) instance_queue.push_back(PlaintextBackend::from_bytes_le(v));
  }
  void ingest_witness(const Witness& w) {
    ingest_header(w.header);
    for (const Value& v : w.short_witness) witness_queue.push_back(PlaintextBackend::from_bytes_le(v));
  }
  void ingest_relation(const Relation& r, PlaintextBackend& backend) {
    ingest_header(r.header);
    is_boolean = (r.gate_mask & M_BOOL) == M_BOOL;
    backend.set_field(r.header.field_characteristic, r.header.field_degree, is_boolean);
    if (!r.gates.empty()) verified_at_least_one_gate = true;
    for (const Function& f : r.functions) {
      FunctionDeclaration d{f.body, (size_t)f.instance_count, (size_t)f.witness_count, (size_t)f.output_count,
                            (size_t)f.input_count};
      known_functions[f.name] = std::move(d);
    }
    Iterators known_iterators;
    Ctx c{&backend, &known_functions, &modulus, is_boolean};
    for (const Gate& g : r.gates) ingest_gate(g, c, values, known_iterators, instance_queue, witness_queue, nullptr);
  }
  // evaluator.rs:213-230
  void ingest_message(const Message& msg, PlaintextBackend& backend) {
    if (has_error) return;
    try {
      switch (msg.kind) {
        case Message::IsInstance: ingest_instance(msg.instance); break;
        case Message::IsWitness: ingest_witness(msg.witness); break;
        case Message::IsRelation: ingest_relation(msg.relation, backend); break;
      }
    } catch (const std::exception& e) {
      has_error = true;
      found_error = e.what();
      if (found_error.rfind("panic:", 0) == 0) panicked = true;
    }
  }
  // evaluator.rs:199-208
  std::vector<std::string> get_violations() const {
    std::vector<std::string> v;
    if (!verified_at_least_one_gate) v.push_back("Did not receive any gate to verify.");
    if (has_error) v.push_back(found_error);
    return v;
  }
};

// consumers/source.rs:69-89: sort by path, then stable by kind (instance<witness<relation<other)
static int file_rank(const std::string& path) {
  std::string name = path.substr(path.find_last_of('/') == std::string::npos ? 0 : path.find_last_of('/') + 1);
  if (name.find("instance") != std::string::npos) return 0;
  if (name.find("witness") != std::string::npos) return 1;
  if (name.find("relation") != std::string::npos) return 3;
  return 4;
}

struct Session {
  Evaluator ev;
  PlaintextBackend backend;
  std::vector<std::string> violations;
  std::string trace_text;
  std::string last_error;
  std::vector<uint8_t> scratch;
};

static void session_ingest_stream(Session& s, const uint8_t* p, size_t n) {
  for (auto& pr : split_messages(p, n)) {
    Message msg;
    try {
      msg = message_from(p + pr.first, pr.second);
    } catch (const std::exception& e) {
      // Evaluator::from_messages unwraps parse errors (evaluator.rs:193): a panic.
      if (!s.ev.has_error) {
        s.ev.has_error = true;
        s.ev.panicked = true;
        s.ev.found_error = std::string("panic: ") + e.what();
      }
      return;
    }
    s.ev.ingest_message(msg, s.backend);
  }
}

}  // namespace zko

using namespace zko;

extern "C" {

void* zko_new(int trace_on) {
  Session* s = new Session();
  s->backend.trace_on = trace_on != 0;
  return s;
}
void zko_free(void* h) { delete (Session*)h; }
void zko_set_max_ops(void* h, uint64_t n) { ((Session*)h)->backend.max_ops = n; }

// A buffer holding one or more size-prefixed messages (Source::from_buffers).
void zko_ingest_buffer(void* h, const uint8_t* p, size_t n) { session_ingest_stream(*(Session*)h, p, n); }

// Files in the order Source::from_filenames would read them.
int zko_ingest_files(void* h, const char** paths, int n) {
  Session& s = *(Session*)h;
  std::vector<std::string> v(paths, paths + n);
  std::sort(v.begin(), v.end());
  std::stable_sort(v.begin(), v.end(), [](const std::string& a, const std::string& b) { return file_rank(a) < file_rank(b); });
  for (const std::string& path : v) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "Warning: failed to open file %s\n", path.c_str()); continue; }
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    session_ingest_stream(s, buf.data(), buf.size());
  }
  return 0;
}

int zko_n_violations(void* h) {
  Session& s = *(Session*)h;
  s.violations = s.ev.get_violations();
  return (int)s.violations.size();
}
const char* zko_violation(void* h, int i) { return ((Session*)h)->violations[i].c_str(); }
int zko_panicked(void* h) { return ((Session*)h)->ev.panicked ? 1 : 0; }
uint64_t zko_n_ops(void* h) { return ((Session*)h)->backend.n_ops; }
uint64_t zko_n_asserts(void* h) { return ((Session*)h)->backend.n_asserts; }
uint64_t zko_trace_len(void* h) { return ((Session*)h)->backend.trace_kind.size(); }
const uint8_t* zko_trace_kinds(void* h) { return ((Session*)h)->backend.trace_kind.data(); }
// "name:value" lines joined by '\n' (SURVEY.md Appendix A digest format)
const char* zko_trace_text(void* h) {
  Session& s = *(Session*)h;
  s.trace_text.clear();
  for (size_t i = 0; i < s.backend.trace_kind.size(); ++i) {
    if (i) s.trace_text.push_back('\n');
    s.trace_text += kTraceNames[s.backend.trace_kind[i]];
    s.trace_text.push_back(':');
    s.trace_text += s.backend.trace_val[i].to_dec();
  }
  return s.trace_text.c_str();
}
// values as fixed-width little-endian; returns number of values that did not fit
uint64_t zko_trace_values_le(void* h, uint8_t* out, uint32_t width) {
  Session& s = *(Session*)h;
  uint64_t bad = 0;
  for (size_t i = 0; i < s.backend.trace_val.size(); ++i)
    if (!s.backend.trace_val[i].to_bytes_le(out + i * width, width)) ++bad;
  return bad;
}
// Evaluator::get (evaluator.rs:750-752): 1 = found
int zko_get_wire_le(void* h, uint64_t id, uint8_t* out, uint32_t width) {
  Session& s = *(Session*)h;
  auto it = s.ev.values.find(id);
  if (it == s.ev.values.end()) return 0;
  return it->second.to_bytes_le(out, width) ? 1 : -1;
}
uint64_t zko_n_live_wires(void* h) { return ((Session*)h)->ev.values.size(); }
uint64_t zko_queue_len(void* h, int which) {
  Session& s = *(Session*)h;
  return which == 0 ? s.ev.instance_queue.size() : s.ev.witness_queue.size();
}

// modexp known-answer hook (evaluator.rs:950-984 test_exponentiation): returns
// base^exponent mod modulus through the same exp() ladder the Switch uses.
int zko_exp(const uint8_t* base, uint32_t bl, const uint8_t* exponent, uint32_t el, const uint8_t* modulus,
            uint32_t ml, uint8_t* out, uint32_t width) {
  try {
    PlaintextBackend backend;
    Value mod(modulus, modulus + ml);
    backend.set_field(mod, 1, false);
    BigUint m = BigUint::from_bytes_le(modulus, ml);
    Functions none;
    Ctx c{&backend, &none, &m, false};
    BigUint r = exp_(c, BigUint::from_bytes_le(base, bl), BigUint::from_bytes_le(exponent, el));
    return r.to_bytes_le(out, width) ? 0 : -1;
  } catch (const std::exception&) {
    return -2;
  }
}

// Batch evaluation used for parity checks at scale and for the CPU baseline:
// `relation` is a stream of size-prefixed Relation messages; lane i gets
// n_inst instance values and n_wit witness values of `width` bytes each
// (little-endian).  One reference Evaluator run per lane, `threads` std::threads.
// first_fail_msg: per-lane violation text is not returned; ok[i] = 1 iff 0 violations.
// Returns wall-clock seconds spent evaluating (excludes relation parsing).
double zko_eval_batch(const uint8_t* relation, size_t relation_len, const uint8_t* header_modulus, uint32_t mod_len,
                      const uint8_t* inst, uint32_t n_inst, const uint8_t* wit, uint32_t n_wit, uint32_t width,
                      uint32_t batch, uint32_t threads, uint8_t* ok, uint64_t* total_ops) {
  std::vector<Message> rel_msgs;
  for (auto& pr : split_messages(relation, relation_len)) rel_msgs.push_back(message_from(relation + pr.first, pr.second));
  Header hdr;
  hdr.version = "1.0.0";
  hdr.field_characteristic.assign(header_modulus, header_modulus + mod_len);
  hdr.field_degree = 1;
  std::atomic<uint32_t> next(0);
  std::atomic<uint64_t> ops(0);
  auto t0 = std::chrono::steady_clock::now();
  auto worker = [&]() {
    for (;;) {
      uint32_t lane = next.fetch_add(1);
      if (lane >= batch) break;
      Evaluator ev;
      PlaintextBackend backend;
      Message mi, mw;
      mi.kind = Message::IsInstance;
      mi.instance.header = hdr;
      for (uint32_t k = 0; k < n_inst; ++k) {
        const uint8_t* p = inst + ((size_t)lane * n_inst + k) * width;
        mi.instance.common_inputs.emplace_back(p, p + width);
      }
      mw.kind = Message::IsWitness;
      mw.witness.header = hdr;
      for (uint32_t k = 0; k < n_wit; ++k) {
        const uint8_t* p = wit + ((size_t)lane * n_wit + k) * width;
        mw.witness.short_witness.emplace_back(p, p + width);
      }
      ev.ingest_message(mi, backend);
      ev.ingest_message(mw, backend);
      for (const Message& m : rel_msgs) ev.ingest_message(m, backend);
      ok[lane] = ev.get_violations().empty() ? 1 : 0;
      ops += backend.n_ops + backend.n_asserts;
    }
  };
  std::vector<std::thread> pool;
  for (uint32_t t = 1; t < threads; ++t) pool.emplace_back(worker);
  worker();
  for (auto& th : pool) th.join();
  auto t1 = std::chrono::steady_clock::now();
  if (total_ops) *total_ops = ops.load();
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
