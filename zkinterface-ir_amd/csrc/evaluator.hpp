// Evaluator<B>: the host-side control flow of the reference's
// `consumers::evaluator::Evaluator<B: ZKBackend>` (rust/src/consumers/evaluator.rs:
// 158-303 state and message ingestion, :318-691 ingest_gate, :698-746
// ingest_subcircuit, :801-839 exp / compute_weight), re-stated in C++ because
// the image has no Rust toolchain.  It inlines functions, unrolls loops and
// multiplexes switches into calls on a ZKBackend `B`; with B = TapeBackend
// those calls are recorded into the linear gate tape the HIP kernels replay.
//
// A backend B provides (names follow the trait at evaluator.rs:17-76):
//   using Wire, FieldElement
//   static FieldElement from_bytes_le(const Value&)
//   void set_field(const Value& modulus, uint32_t degree, bool is_boolean)
//   FieldElement one() / minus_one() / zero()
//   Wire copy(w) constant(fe) add(a,b) multiply(a,b) add_constant(a,fe)
//        mul_constant(a,fe) and_(a,b) xor_(a,b) not_(a) instance(fe) witness(fe*)
//   void assert_zero(w)            -- throws zki::Error when it can tell the wire is non-zero
//   void note_assert_wire(WireId)  -- optional: local id of the wire about to be asserted
//   size_t note_ladder_begin(); void note_ladder_end(size_t, Wire base, Wire result)  -- optional: the calls in
//     between computed result = base^(modulus - 1) (the Switch indicator, evaluator.rs:801-839)
#pragma once
#include <algorithm>
#include <deque>
#include <string>
#include <unordered_map>
#include <vector>

#include "sieve/reader.hpp"
#include "sieve/structs.hpp"

namespace zki {

// scope: HashMap<WireId, B::Wire>.  Wire ids are small and dense in practice,
// so ids below a bound live in a flat vector; the rest fall back to a hash map.
template <class W>
class WireScope {
 public:
  // HashMap::insert: stores the value, returns true if the key was occupied.
  bool insert(WireId id, W w) {
    if (id < kDenseLimit) {
      if (id >= dense_.size()) {
        const size_t n = std::max<size_t>(id + 1, dense_.size() * 2);
        dense_.resize(n);
        present_.resize(n, 0);
      }
      const bool had = present_[id];
      dense_[id] = std::move(w);
      present_[id] = 1;
      return had;
    }
    auto r = sparse_.insert_or_assign(id, std::move(w));
    return !r.second;
  }
  const W* find(WireId id) const {
    if (id < kDenseLimit) return (id < dense_.size() && present_[id]) ? &dense_[id] : nullptr;
    auto it = sparse_.find(id);
    return it == sparse_.end() ? nullptr : &it->second;
  }
  bool erase(WireId id) {
    if (id < kDenseLimit) {
      if (id >= dense_.size() || !present_[id]) return false;
      present_[id] = 0;
      dense_[id] = W();
      return true;
    }
    return sparse_.erase(id) != 0;
  }
  template <class F>
  void for_each(F&& f) const {
    for (size_t i = 0; i < dense_.size(); ++i)
      if (present_[i]) f((WireId)i, dense_[i]);
    for (const auto& kv : sparse_) f(kv.first, kv.second);
  }
  template <class F>
  void for_each_mut(F&& f) {   // the same, the wires may be replaced (not the keys)
    for (size_t i = 0; i < dense_.size(); ++i)
      if (present_[i]) f((WireId)i, dense_[i]);
    for (auto& kv : sparse_) f(kv.first, kv.second);
  }

 private:
  static constexpr WireId kDenseLimit = 1ull << 24;
  std::vector<W> dense_;
  std::vector<uint8_t> present_;
  std::unordered_map<WireId, W> sparse_;
};

template <class B>
class Evaluator {
 public:
  using Wire = typename B::Wire;
  using FieldElement = typename B::FieldElement;
  using Scope = WireScope<Wire>;
  using Queue = std::deque<FieldElement>;

  // evaluator.rs:191-195.  Parse errors are panics in the reference (`msg.unwrap()`).
  static Evaluator from_messages(const Source& source, B& backend) {
    Evaluator ev;
    source.for_each_buffer([&](const uint8_t* p, size_t n) { ev.ingest_buffer(p, n, backend); });
    return ev;
  }

  // one size-prefixed message
  void ingest_buffer(const uint8_t* p, size_t n, B& backend) {
    if (found_error_) return;
    Message msg;
    try {
      msg = read_message(p, n);
    } catch (const std::exception& e) {
      latch(std::string("panic: ") + e.what(), true);
      return;
    }
    ingest_message(msg, backend);
  }

  // evaluator.rs:199-208
  std::vector<std::string> get_violations() const {
    std::vector<std::string> v;
    if (!verified_at_least_one_gate_) v.push_back("Did not receive any gate to verify.");
    if (found_error_) v.push_back(error_);
    return v;
  }

  // evaluator.rs:213-230
  void ingest_message(const Message& msg, B& backend) {
    if (found_error_) return;
    try {
      switch (msg.kind) {
        case Message::IsInstance: ingest_instance(msg.instance); break;
        case Message::IsWitness: ingest_witness(msg.witness); break;
        case Message::IsRelation: ingest_relation(msg.relation, backend); break;
        default: break;
      }
    } catch (const Panic& e) {
      latch(std::string("panic: ") + e.what(), true);
    } catch (const std::exception& e) {
      latch(e.what(), false);
    }
  }

  // evaluator.rs:239-257
  void ingest_instance(const Instance& instance) {
    ingest_header(instance.header);
    for (const Value& v : instance.common_inputs) instance_queue_.push_back(B::from_bytes_le(v));
  }
  void ingest_witness(const Witness& witness) {
    ingest_header(witness.header);
    for (const Value& v : witness.short_witness) witness_queue_.push_back(B::from_bytes_le(v));
  }
  // Batch extension: the queues hold whatever FieldElements the backend wants
  // to see again in instance()/witness() (for TapeBackend: stream positions).
  void push_instance(FieldElement fe) { instance_queue_.push_back(std::move(fe)); }
  void push_witness(FieldElement fe) { witness_queue_.push_back(std::move(fe)); }
  void set_modulus(const Value& m) { modulus_ = m; }

  // evaluator.rs:260-303
  void ingest_relation(const Relation& relation, B& backend) {
    ingest_header(relation.header);
    is_boolean_ = mask::contains_feature(relation.gate_mask, mask::BOOL);
    backend.set_field(relation.header.field_characteristic, relation.header.field_degree, is_boolean_);
    if (!relation.gates.empty()) verified_at_least_one_gate_ = true;
    for (const Function& f : relation.functions) {
      FunctionDeclaration d;
      d.subcircuit = f.body;
      d.instance_nbr = f.instance_count;
      d.witness_nbr = f.witness_count;
      d.output_count = f.output_count;
      d.input_count = f.input_count;
      known_functions_[f.name] = std::move(d);
    }
    IteratorScope known_iterators;
    Env env{backend, known_functions_, exponent_of(modulus_), is_boolean_};
    for (const Gate& gate : relation.gates)
      ingest_gate(gate, env, values_, known_iterators, instance_queue_, witness_queue_, nullptr);
  }

  // evaluator.rs:750-752
  const Wire* get(WireId id) const { return values_.find(id); }
  const Scope& values() const { return values_; }
  // the top-level scope, wires replaceable: a backend that opens a new field segment re-binds the wires that live on
  // (between messages no other scope exists)
  Scope& values_mut() { return values_; }

  bool has_error() const { return found_error_; }
  bool panicked() const { return panicked_; }
  const std::string& error() const { return error_; }
  size_t instance_queue_len() const { return instance_queue_.size(); }
  size_t witness_queue_len() const { return witness_queue_.size(); }
  bool is_boolean() const { return is_boolean_; }

 private:
  struct FunctionDeclaration {  // evaluator.rs:130-136
    std::shared_ptr<Subcircuit> subcircuit;
    uint64_t instance_nbr = 0, witness_nbr = 0, output_count = 0, input_count = 0;
  };
  using Functions = std::unordered_map<std::string, FunctionDeclaration>;

  // modulus - 1 as a little-endian bit string: the exponent of the Switch
  // indicator (evaluator.rs:832); only its bits are needed by exp().
  struct Exponent {
    std::vector<uint8_t> le;  // p - 1
    size_t top_bit = 0;       // index of the most significant set bit
    bool zero = true;
    bool bit(size_t i) const { return (le[i / 8] >> (i % 8)) & 1; }
  };
  static Exponent exponent_of(const Value& modulus) {
    Exponent e;
    e.le = modulus;
    // BigUint subtraction: p - 1 (panics on p == 0 in the reference; set_field rejects 0 first)
    size_t i = 0;
    while (i < e.le.size() && e.le[i] == 0) e.le[i++] = 0xff;
    if (i < e.le.size()) e.le[i] -= 1;
    for (size_t b = e.le.size() * 8; b-- > 0;)
      if (e.bit(b)) {
        e.top_bit = b;
        e.zero = false;
        break;
      }
    return e;
  }

  struct Env {
    B& backend;
    const Functions& known_functions;
    Exponent exponent;
    bool is_boolean;
  };

  void latch(const std::string& msg, bool panic) {
    found_error_ = true;
    panicked_ = panic;
    error_ = msg;
  }
  void ingest_header(const Header& h) { modulus_ = h.field_characteristic; }  // evaluator.rs:232-235

  // evaluator.rs:775-797
  static void set(Scope& scope, WireId id, Wire w) {
    if (scope.insert(id, std::move(w))) throw Error("Wire_" + std::to_string(id) + " already has a value in this scope.");
  }
  static const Wire& get(const Scope& scope, WireId id) {
    const Wire* w = scope.find(id);
    if (!w) throw Error("No value given for wire_" + std::to_string(id));
    return *w;
  }
  static void remove(Scope& scope, WireId id) {
    if (!scope.erase(id)) throw Error("No value given for wire_" + std::to_string(id));
  }

  // evaluator.rs:80-126
  static Wire as_mul(Env& e, const Wire& a, const Wire& b) { return e.is_boolean ? e.backend.and_(a, b) : e.backend.multiply(a, b); }
  static Wire as_add(Env& e, const Wire& a, const Wire& b) { return e.is_boolean ? e.backend.xor_(a, b) : e.backend.add(a, b); }
  static Wire as_negate(Env& e, const Wire& w) { return e.is_boolean ? e.backend.copy(w) : e.backend.mul_constant(w, e.backend.minus_one()); }
  static Wire as_add_one(Env& e, const Wire& w) { return e.is_boolean ? e.backend.not_(w) : e.backend.add_constant(w, e.backend.one()); }

  // evaluator.rs:801-820: recursive square-and-multiply on exponent >> k.
  // exp(e >> k) for k = top_bit is `copy(base)`; unwinding multiplies on the way back.
  static Wire exp(Env& e, const Wire& base, size_t shift) {
    const Exponent& x = e.exponent;
    if (shift == x.top_bit) return e.backend.copy(base);  // exponent.is_one()
    Wire previous = exp(e, base, shift + 1);
    Wire ret = as_mul(e, previous, previous);
    if (x.bit(shift)) return as_mul(e, ret, base);
    return ret;
  }
  // evaluator.rs:823-839
  static Wire compute_weight(Env& e, const Value& case_, const Wire& condition) {
    Wire case_wire = e.backend.constant(B::from_bytes_le(case_));
    Wire minus_cond = as_negate(e, condition);
    Wire base = as_add(e, case_wire, minus_cond);
    if (e.exponent.zero) throw Panic("exponent 0 never reaches 1 (modulus 1)");  // unbounded recursion in the reference
    const size_t mark = ladder_begin(e.backend, 0);
    Wire base_to_exp = exp(e, base, 0);
    ladder_end(e.backend, mark, base, base_to_exp, 0);  // base_to_exp = base^(modulus - 1) -- by `and`s under is_boolean: the backend's call what to make of it
    Wire right = as_negate(e, base_to_exp);
    return as_add_one(e, right);
  }

  static std::string wrong_count(const char* what, const std::string& name, uint64_t expected, size_t got) {
    return std::string("Wrong number of ") + what + " variables in call to function " + name + " (Expected " +
           std::to_string(expected) + " / Got " + std::to_string(got) + ").";
  }

  // optional backend hooks around the exponent ladder of a Switch weight: a recording backend learns which of its
  // calls compute base^(modulus - 1) and may evaluate that differently when it replays them
  template <class BB>
  static auto ladder_begin(BB& b, int) -> decltype(b.note_ladder_begin()) { return b.note_ladder_begin(); }
  template <class BB>
  static size_t ladder_begin(BB&, long) { return 0; }
  template <class BB>
  static auto ladder_end(BB& b, size_t mark, const Wire& base, const Wire& result, int)
      -> decltype(b.note_ladder_end(mark, base, result), void()) { b.note_ladder_end(mark, base, result); }
  template <class BB>
  static void ladder_end(BB&, size_t, const Wire&, const Wire&, long) {}

  template <class BB>
  static auto note_assert(BB& b, WireId id, int) -> decltype(b.note_assert_wire(id), void()) { b.note_assert_wire(id); }
  template <class BB>
  static void note_assert(BB&, WireId, long) {}

  // evaluator.rs:698-746
  static void ingest_subcircuit(const Subcircuit& subcircuit, Env& e, const std::vector<WireId>& output_list,
                                const std::vector<WireId>& input_list, Scope& scope, IteratorScope& known_iterators,
                                Queue& instances, Queue& witnesses, const Wire* weight) {
    // a function that (indirectly) calls itself would recurse until the stack is gone (it does in the
    // reference); refuse at a depth no legitimate relation reaches
    struct Depth {
      explicit Depth(int* d) : d_(d) {
        if (++*d_ > 2000) {
          --*d_;
          throw Error("subcircuits nested deeper than 2000 calls");
        }
      }
      ~Depth() { --*d_; }
      int* d_;
    };
    static thread_local int depth = 0;
    Depth guard(&depth);
    Scope new_scope;
    for (size_t idx = 0; idx < input_list.size(); ++idx) {
      const Wire& i = get(scope, input_list[idx]);
      set(new_scope, (WireId)(idx + output_list.size()), e.backend.copy(i));
    }
    for (const Gate& g : subcircuit) ingest_gate(g, e, new_scope, known_iterators, instances, witnesses, weight);
    for (size_t idx = 0; idx < output_list.size(); ++idx) {
      const Wire& w = get(new_scope, (WireId)idx);
      set(scope, output_list[idx], e.backend.copy(w));
    }
  }

  // evaluator.rs:318-691
  static void ingest_gate(const Gate& gate, Env& e, Scope& scope, IteratorScope& known_iterators, Queue& instances,
                          Queue& witnesses, const Wire* weight) {
    B& backend = e.backend;
    switch (gate.kind) {
      case GateKind::Constant: {
        Wire wire = backend.constant(B::from_bytes_le(gate.ext->constant));
        set(scope, gate.out, std::move(wire));
        break;
      }
      case GateKind::AssertZero: {
        const Wire& inp_wire = get(scope, gate.in0);
        Wire should_be_zero = weight ? as_mul(e, *weight, inp_wire) : backend.copy(inp_wire);
        note_assert(backend, gate.in0, 0);
        try {
          backend.assert_zero(should_be_zero);
        } catch (const Error&) {
          throw Error("Wire_" + std::to_string(gate.in0) + " (may be weighted) should be 0, while it is not");
        }
        break;
      }
      case GateKind::Copy: {
        Wire out_wire = backend.copy(get(scope, gate.in0));
        set(scope, gate.out, std::move(out_wire));
        break;
      }
      case GateKind::Add: {
        const Wire& l = get(scope, gate.in0);
        const Wire& r = get(scope, gate.in1);
        set(scope, gate.out, backend.add(l, r));
        break;
      }
      case GateKind::Mul: {
        const Wire& l = get(scope, gate.in0);
        const Wire& r = get(scope, gate.in1);
        set(scope, gate.out, backend.multiply(l, r));
        break;
      }
      case GateKind::AddConstant: {
        const Wire& l = get(scope, gate.in0);
        FieldElement r = B::from_bytes_le(gate.ext->constant);
        set(scope, gate.out, backend.add_constant(l, std::move(r)));
        break;
      }
      case GateKind::MulConstant: {
        const Wire& l = get(scope, gate.in0);
        FieldElement r = B::from_bytes_le(gate.ext->constant);
        set(scope, gate.out, backend.mul_constant(l, std::move(r)));
        break;
      }
      case GateKind::And: {
        const Wire& l = get(scope, gate.in0);
        const Wire& r = get(scope, gate.in1);
        set(scope, gate.out, backend.and_(l, r));
        break;
      }
      case GateKind::Xor: {
        const Wire& l = get(scope, gate.in0);
        const Wire& r = get(scope, gate.in1);
        set(scope, gate.out, backend.xor_(l, r));
        break;
      }
      case GateKind::Not: {
        const Wire& v = get(scope, gate.in0);
        set(scope, gate.out, backend.not_(v));
        break;
      }
      case GateKind::Instance: {
        if (instances.empty()) throw Error("Not enough instance to consume");
        FieldElement val = std::move(instances.front());
        instances.pop_front();
        set(scope, gate.out, backend.instance(std::move(val)));
        break;
      }
      case GateKind::Witness: {
        if (witnesses.empty()) {
          set(scope, gate.out, backend.witness(nullptr));
        } else {
          FieldElement val = std::move(witnesses.front());
          witnesses.pop_front();
          set(scope, gate.out, backend.witness(&val));
        }
        break;
      }
      case GateKind::Free: {
        const WireId last = gate.has_last ? gate.in1 : gate.in0;
        for (WireId cur = gate.in0; cur <= last; ++cur) {
          remove(scope, cur);
          if (cur == UINT64_MAX) break;
        }
        break;
      }
      case GateKind::Call: {
        const GateExt& x = *gate.ext;
        auto it = e.known_functions.find(x.name);
        if (it == e.known_functions.end()) throw Error("Unknown function");
        const FunctionDeclaration& f = it->second;
        const std::vector<WireId> expanded_output = expand_wirelist(x.output_wires);
        const std::vector<WireId> expanded_input = expand_wirelist(x.input_wires);
        if (expanded_output.size() != f.output_count)
          throw Error(wrong_count("output", x.name, f.output_count, expanded_output.size()));
        if (expanded_input.size() != f.input_count)
          throw Error(wrong_count("input", x.name, f.input_count, expanded_input.size()));
        IteratorScope no_iterators;  // named calls do not see the caller's iterators
        ingest_subcircuit(*f.subcircuit, e, expanded_output, expanded_input, scope, no_iterators, instances,
                          witnesses, weight);
        break;
      }
      case GateKind::AnonCall: {
        const GateExt& x = *gate.ext;
        const std::vector<WireId> expanded_output = expand_wirelist(x.output_wires);
        const std::vector<WireId> expanded_input = expand_wirelist(x.input_wires);
        ingest_subcircuit(*x.subcircuit, e, expanded_output, expanded_input, scope, known_iterators, instances,
                          witnesses, weight);
        break;
      }
      case GateKind::For: {
        const GateExt& x = *gate.ext;
        const ForLoopBody& body = x.body;
        for (uint64_t i = x.first; i <= x.last; ++i) {
          known_iterators.insert(x.name, i);
          const std::vector<WireId> expanded_output = evaluate_iterexpr_list(body.outputs, known_iterators);
          if (!body.anonymous) {
            auto it = e.known_functions.find(body.name);
            if (it == e.known_functions.end()) throw Error("Unknown function");
            const FunctionDeclaration& f = it->second;
            const std::vector<WireId> expanded_input = evaluate_iterexpr_list(body.inputs, known_iterators);
            if (expanded_output.size() != f.output_count)
              throw Error(wrong_count("output", body.name, f.output_count, expanded_output.size()));
            if (expanded_input.size() != f.input_count)
              throw Error(wrong_count("input", body.name, f.input_count, expanded_input.size()));
            IteratorScope no_iterators;
            ingest_subcircuit(*f.subcircuit, e, expanded_output, expanded_input, scope, no_iterators, instances,
                              witnesses, weight);
          } else {
            const std::vector<WireId> expanded_input = evaluate_iterexpr_list(body.inputs, known_iterators);
            ingest_subcircuit(*body.subcircuit, e, expanded_output, expanded_input, scope, known_iterators, instances,
                              witnesses, weight);
          }
          if (i == UINT64_MAX) break;
        }
        known_iterators.remove(x.name);
        break;
      }
      case GateKind::Switch: ingest_switch(gate, e, scope, known_iterators, instances, witnesses, weight); break;
      default: throw Error("No gate type");
    }
  }

  // evaluator.rs:563-688
  static void ingest_switch(const Gate& gate, Env& e, Scope& scope, IteratorScope& known_iterators, Queue& instances,
                            Queue& witnesses, const Wire* weight) {
    B& backend = e.backend;
    const GateExt& x = *gate.ext;
    uint64_t max_instance_count = 0, max_witness_count = 0;
    for (const CaseInvoke& branch : x.branches) {
      uint64_t ic, wc;
      if (!branch.anonymous) {
        auto it = e.known_functions.find(branch.name);
        if (it == e.known_functions.end()) throw Error("Unknown function");
        ic = it->second.instance_nbr;
        wc = it->second.witness_nbr;
      } else {
        ic = branch.instance_count;
        wc = branch.witness_count;
      }
      max_instance_count = std::max(max_instance_count, ic);
      max_witness_count = std::max(max_witness_count, wc);
    }
    // split_off + swap (:586-591): the first min(len, max) queued values are
    // taken out once; every branch starts from its own copy of them.
    auto take_front = [](Queue& q, uint64_t n) {
      const size_t k = (size_t)std::min<uint64_t>(q.size(), n);
      Queue head(q.begin(), q.begin() + k);
      q.erase(q.begin(), q.begin() + k);
      return head;
    };
    const Queue new_instances = take_front(instances, max_instance_count);
    const Queue new_witnesses = take_front(witnesses, max_witness_count);

    std::vector<Scope> branches_scope;
    const std::vector<WireId> expanded_output = expand_wirelist(x.output_wires);
    std::vector<Wire> weights;
    const size_t n_branches = std::min(x.cases.size(), x.branches.size());  // zip
    for (size_t k = 0; k < n_branches; ++k) {
      const CaseInvoke& branch = x.branches[k];
      Wire branch_weight = compute_weight(e, x.cases[k], get(scope, gate.in0));
      Wire weighted_branch_weight = weight ? as_mul(e, *weight, branch_weight) : std::move(branch_weight);
      Scope branch_scope;
      Queue branch_instances = new_instances, branch_witnesses = new_witnesses;
      if (!branch.anonymous) {
        auto it = e.known_functions.find(branch.name);
        if (it == e.known_functions.end()) throw Error("Unknown function: " + branch.name);
        const FunctionDeclaration& f = it->second;
        const std::vector<WireId> expanded_input = expand_wirelist(branch.input_wires);
        if (expanded_output.size() != f.output_count)
          throw Error(wrong_count("output", branch.name, f.output_count, expanded_output.size()));
        if (expanded_input.size() != f.input_count)
          throw Error(wrong_count("input", branch.name, f.input_count, expanded_input.size()));
        for (WireId wire : expanded_input) branch_scope.insert(wire, backend.copy(get(scope, wire)));
        IteratorScope no_iterators;
        ingest_subcircuit(*f.subcircuit, e, expanded_output, expanded_input, branch_scope, no_iterators,
                          branch_instances, branch_witnesses, &weighted_branch_weight);
      } else {
        const std::vector<WireId> expanded_input = expand_wirelist(branch.input_wires);
        for (WireId wire : expanded_input) branch_scope.insert(wire, backend.copy(get(scope, wire)));
        ingest_subcircuit(*branch.subcircuit, e, expanded_output, expanded_input, branch_scope, known_iterators,
                          branch_instances, branch_witnesses, &weighted_branch_weight);
      }
      weights.push_back(std::move(weighted_branch_weight));
      branches_scope.push_back(std::move(branch_scope));
    }
    // weighted sum of every output over the branches (:673-687)
    for (WireId output_wire : expanded_output) {
      Wire accu = backend.constant(backend.zero());
      for (size_t k = 0; k < branches_scope.size(); ++k) {
        Wire weighted_wire = as_mul(e, get(branches_scope[k], output_wire), weights[k]);
        accu = as_add(e, accu, weighted_wire);
      }
      set(scope, output_wire, std::move(accu));
    }
  }

  Scope values_;
  Value modulus_;
  Queue instance_queue_, witness_queue_;
  bool is_boolean_ = false;
  Functions known_functions_;
  bool verified_at_least_one_gate_ = false;
  bool found_error_ = false, panicked_ = false;
  std::string error_;
};

}  // namespace zki
