// Validator: the specification checks `zki_sieve validate` / `valid-eval-metrics` run beside the
// Evaluator (rust/src/consumers/validator.rs:64-861; CLI rust/src/cli.rs:302-313,333-363).
// Same state, same order of checks and the same violation strings as the reference, so that the
// three-part report of `valid-eval-metrics` can be reproduced line for line.
#pragma once
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "sieve/bignum.hpp"
#include "sieve/reader.hpp"
#include "sieve/structs.hpp"

namespace zki {

class Validator {
 public:
  static Validator new_as_verifier() { return Validator(); }  // validator.rs:108-110
  static Validator new_as_prover() {                           // validator.rs:112-117
    Validator v;
    v.as_prover_ = true;
    return v;
  }
  Validator();

  void ingest_message(const Message& msg);     // :152-158
  void ingest_instance(const Instance& i);     // :200-209
  void ingest_witness(const Witness& w);       // :211-223
  void ingest_relation(const Relation& r);     // :225-288

  // get_violations(self) consumes the validator in the reference (:135-142); here it works on a
  // copy of the counters so that it can be asked more than once.
  std::vector<std::string> get_violations() const;
  const std::vector<std::string>& get_strict_violations() const { return violations_; }  // :144-146
  size_t how_many_violations() const { return violations_.size(); }                       // :148-150
  bool has_live_wires() const { return !live_wires_.empty(); }  // "WARNING: few variables were not freed."

  // Free / For ranges are walked wire by wire like the reference does; a bound on the total number
  // of steps keeps a corrupt range from running for ever (throws zki::Error when exceeded).
  void set_max_steps(uint64_t n) { *steps_left_ = n; }

  static const char* implemented_checks();  // :26-62, printed by `list-validations`

 private:
  using FunctionTable = std::map<std::string, std::array<uint64_t, 4>>;  // out, in, instance, witness counts

  void ingest_header(const Header& h);         // :160-198
  void ingest_gate(const Gate& g);             // :290-642
  bool ingest_call(const std::string& name, size_t n_out, size_t n_in, uint64_t* ins, uint64_t* wit);  // :649-672
  void ingest_subcircuit(const Subcircuit& sub, uint64_t output_count, uint64_t input_count,
                         uint64_t instance_count, uint64_t witness_count, bool use_same_scope);        // :683-737

  bool is_defined(WireId id) const { return live_wires_.count(id) != 0; }
  void declare(WireId id) { live_wires_.insert(id); }
  void remove(WireId id);
  void consume_instance(uint64_t n);
  void consume_witness(uint64_t n);
  void ensure_defined_and_set(WireId id);
  void ensure_undefined(WireId id);
  void ensure_undefined_and_set(WireId id);
  void ensure_value_in_field(const Value& v, const std::string& name);
  void ensure_allowed_gate(const char* name, uint16_t mask_bit);
  void ensure_allowed_feature(const char* name, uint16_t mask_bit);
  void violate(const std::string& msg) { violations_.push_back(msg); }
  std::vector<WireId> expand_or_violate(const WireList& l);
  void step(uint64_t n = 1);

  bool as_prover_ = false;
  uint64_t instance_queue_len_ = 0, witness_queue_len_ = 0;
  std::set<WireId> live_wires_;
  bool got_header_ = false;
  uint16_t gate_set_ = 0, features_ = 0;
  std::string header_version_;
  BigNat field_characteristic_;
  uint64_t field_degree_ = 0;
  std::shared_ptr<FunctionTable> known_functions_;
  std::shared_ptr<IteratorScope> known_iterators_;
  std::shared_ptr<uint64_t> steps_left_;
  std::vector<std::string> violations_;
};

// The two patterns of validator.rs:22-25, matched the way the `regex` crate does (anchored, `.` = any
// character but '\n').  \d and \w are ASCII here plus, for \w, letters outside ASCII by code-point block.
bool matches_version_pattern(const std::string& s);  // ^\d+.\d+.\d+$
bool matches_name_pattern(const std::string& s);     // ^[a-zA-Z_][\w]*(?:(?:\.|:{2})[a-zA-Z_][\w]*)*$
extern const char* const kNamesRegexText;

}  // namespace zki
