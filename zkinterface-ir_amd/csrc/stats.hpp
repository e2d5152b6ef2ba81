// Stats: the metrics consumer of `zki_sieve metrics` / `valid-eval-metrics`
// (rust/src/consumers/stats.rs:11-287; CLI rust/src/cli.rs:322-330,333-363).  Counts gates with
// function bodies, loop iterations and switch branches expanded, exactly as the reference counts them.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "sieve/structs.hpp"

namespace zki {

struct GateStats {  // stats.rs:12-41, same field order (it is the JSON order)
  uint64_t instance_variables = 0, witness_variables = 0;
  uint64_t constants_gates = 0, assert_zero_gates = 0, copy_gates = 0, add_gates = 0, mul_gates = 0;
  uint64_t add_constant_gates = 0, mul_constant_gates = 0, and_gates = 0, xor_gates = 0, not_gates = 0;
  uint64_t variables_freed = 0;
  uint64_t functions_defined = 0, functions_called = 0;
  uint64_t switches = 0, branches = 0;
  uint64_t for_loops = 0;
  uint64_t instance_messages = 0, witness_messages = 0, relation_messages = 0;

  static constexpr int kFields = 21;
  static const char* field_name(int i);
  uint64_t field(int i) const;
};

struct FunctionStats {
  GateStats stats;
  uint64_t instance_count = 0, witness_count = 0;
};

class Stats {  // stats.rs:44-109
 public:
  Value field_characteristic;
  uint32_t field_degree = 0;
  GateStats gate_stats;
  std::map<std::string, FunctionStats> functions;
  std::vector<std::string> warnings;  // "WARNING Stats: function not defined \"{}\"" (stderr in the reference)

  void ingest_message(const Message& msg);
  void ingest_instance(const Instance& i);
  void ingest_witness(const Witness& w);
  void ingest_relation(const Relation& r);

  // serde_json::to_writer_pretty(&stats) (cli.rs:327,353): two-space indent, one array element per
  // line; the `functions` map is written in name order (the reference's HashMap order is arbitrary).
  std::string to_json_pretty() const;

 private:
  void ingest_header(const Header& h);
  GateStats ingest_subcircuit(const Subcircuit& sub);
  void ingest_gate(GateStats& into, const Gate& g);
  void ingest_named_call(GateStats& into, const std::string& name, uint64_t* ins, uint64_t* wit);
};

}  // namespace zki
