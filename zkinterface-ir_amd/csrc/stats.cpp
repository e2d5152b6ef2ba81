#include "stats.hpp"

#include <algorithm>
#include <array>

namespace zki {

namespace {
const char* const kNames[GateStats::kFields] = {
    "instance_variables", "witness_variables", "constants_gates", "assert_zero_gates", "copy_gates", "add_gates",
    "mul_gates", "add_constant_gates", "mul_constant_gates", "and_gates", "xor_gates", "not_gates", "variables_freed",
    "functions_defined", "functions_called", "switches", "branches", "for_loops", "instance_messages",
    "witness_messages", "relation_messages"};

// The counters ingest_call_stats (stats.rs:268-286) carries from a callee to its caller: gate and structure
// counts; variables, definitions and message counts are not carried.
std::array<uint64_t*, 15> call_fields(GateStats& g) {
  return {&g.constants_gates, &g.assert_zero_gates, &g.copy_gates, &g.add_gates, &g.mul_gates, &g.add_constant_gates,
          &g.mul_constant_gates, &g.and_gates, &g.xor_gates, &g.not_gates, &g.variables_freed, &g.switches,
          &g.branches, &g.for_loops, &g.functions_called};
}
void add_call_stats(GateStats& a, const GateStats& o) {
  GateStats from = o;
  const auto dst = call_fields(a), src = call_fields(from);
  for (size_t k = 0; k < dst.size(); ++k) *dst[k] += *src[k];
}

std::string json_string(const std::string& s) {
  std::string out = "\"";
  for (unsigned char c : s) {
    switch (c) {
      case '"': out += "\\\""; break;
      case '\\': out += "\\\\"; break;
      case '\n': out += "\\n"; break;
      case '\r': out += "\\r"; break;
      case '\t': out += "\\t"; break;
      case '\b': out += "\\b"; break;
      case '\f': out += "\\f"; break;
      default:
        if (c < 0x20) {
          static const char* hex = "0123456789abcdef";
          out += "\\u00";
          out.push_back(hex[c >> 4]);
          out.push_back(hex[c & 15]);
        } else {
          out.push_back((char)c);
        }
    }
  }
  return out + "\"";
}

void write_gate_stats(std::string& out, const GateStats& g, const std::string& indent) {
  out += "{\n";
  for (int i = 0; i < GateStats::kFields; ++i) {
    out += indent + "  \"" + kNames[i] + "\": " + std::to_string(g.field(i));
    out += i + 1 < GateStats::kFields ? ",\n" : "\n";
  }
  out += indent + "}";
}
}  // namespace

const char* GateStats::field_name(int i) { return kNames[i]; }

uint64_t GateStats::field(int i) const {
  const uint64_t v[kFields] = {instance_variables, witness_variables, constants_gates, assert_zero_gates, copy_gates,
                               add_gates, mul_gates, add_constant_gates, mul_constant_gates, and_gates, xor_gates,
                               not_gates, variables_freed, functions_defined, functions_called, switches, branches,
                               for_loops, instance_messages, witness_messages, relation_messages};
  return v[i];
}

void Stats::ingest_message(const Message& msg) {
  switch (msg.kind) {
    case Message::IsInstance: ingest_instance(msg.instance); break;
    case Message::IsWitness: ingest_witness(msg.witness); break;
    case Message::IsRelation: ingest_relation(msg.relation); break;
    default: break;
  }
}

void Stats::ingest_header(const Header& h) {
  field_characteristic = h.field_characteristic;
  field_degree = h.field_degree;
}

void Stats::ingest_instance(const Instance& i) {
  ingest_header(i.header);
  gate_stats.instance_messages += 1;
}

void Stats::ingest_witness(const Witness& w) {
  ingest_header(w.header);
  gate_stats.witness_messages += 1;
}

void Stats::ingest_relation(const Relation& r) {
  ingest_header(r.header);
  gate_stats.relation_messages += 1;
  static const Subcircuit kEmpty;
  for (const Function& f : r.functions) {
    gate_stats.functions_defined += 1;
    FunctionStats fs;
    fs.stats = ingest_subcircuit(f.body ? *f.body : kEmpty);
    fs.instance_count = f.instance_count;
    fs.witness_count = f.witness_count;
    functions[f.name] = fs;
  }
  for (const Gate& g : r.gates) ingest_gate(gate_stats, g);
}

GateStats Stats::ingest_subcircuit(const Subcircuit& sub) {
  GateStats local;
  for (const Gate& g : sub) ingest_gate(local, g);
  return local;
}

void Stats::ingest_named_call(GateStats& into, const std::string& name, uint64_t* ins, uint64_t* wit) {
  into.functions_called += 1;
  *ins = *wit = 0;
  const auto it = functions.find(name);
  if (it == functions.end()) {
    warnings.push_back("WARNING Stats: function not defined \"" + name + "\"");
    return;
  }
  add_call_stats(into, it->second.stats);
  *ins = it->second.instance_count;
  *wit = it->second.witness_count;
}

void Stats::ingest_gate(GateStats& s, const Gate& g) {
  static const Subcircuit kEmpty;
  switch (g.kind) {
    case GateKind::Constant: s.constants_gates += 1; break;
    case GateKind::AssertZero: s.assert_zero_gates += 1; break;
    case GateKind::Copy: s.copy_gates += 1; break;
    case GateKind::Add: s.add_gates += 1; break;
    case GateKind::Mul: s.mul_gates += 1; break;
    case GateKind::AddConstant: s.add_constant_gates += 1; break;
    case GateKind::MulConstant: s.mul_constant_gates += 1; break;
    case GateKind::And: s.and_gates += 1; break;
    case GateKind::Xor: s.xor_gates += 1; break;
    case GateKind::Not: s.not_gates += 1; break;
    case GateKind::Instance: s.instance_variables += 1; break;
    case GateKind::Witness: s.witness_variables += 1; break;
    case GateKind::Free: {
      const WireId last = g.has_last ? g.in1 : g.in0;
      s.variables_freed += last - g.in0 + 1;  // wraps like the reference's release build when last < first
      break;
    }
    case GateKind::Call: {
      uint64_t ins, wit;
      ingest_named_call(s, g.ext->name, &ins, &wit);
      s.instance_variables += ins;
      s.witness_variables += wit;
      break;
    }
    case GateKind::AnonCall: {
      add_call_stats(s, ingest_subcircuit(g.ext->subcircuit ? *g.ext->subcircuit : kEmpty));
      s.instance_variables += g.ext->instance_count;
      s.witness_variables += g.ext->witness_count;
      break;
    }
    case GateKind::Switch: {
      s.switches += 1;
      s.branches += g.ext->branches.size();
      uint64_t max_ins = 0, max_wit = 0;
      for (const CaseInvoke& br : g.ext->branches) {
        uint64_t ins, wit;
        if (!br.anonymous) {
          ingest_named_call(s, br.name, &ins, &wit);
        } else {
          add_call_stats(s, ingest_subcircuit(br.subcircuit ? *br.subcircuit : kEmpty));
          ins = br.instance_count;
          wit = br.witness_count;
        }
        max_ins = std::max(max_ins, ins);
        max_wit = std::max(max_wit, wit);
      }
      s.instance_variables += max_ins;
      s.witness_variables += max_wit;
      break;
    }
    case GateKind::For: {
      s.for_loops += 1;
      const GateExt& x = *g.ext;
      if (x.last < x.first) break;
      // Every iteration adds the same amounts: count one and multiply (the reference loops, and repeats
      // its "function not defined" warning per iteration; here it is recorded once per For gate).
      const ForLoopBody& body = x.body;
      GateStats once;
      uint64_t ins, wit;
      if (!body.anonymous) {
        ingest_named_call(once, body.name, &ins, &wit);
      } else {
        add_call_stats(once, ingest_subcircuit(body.subcircuit ? *body.subcircuit : kEmpty));
        ins = body.instance_count;
        wit = body.witness_count;
      }
      const uint64_t n = x.last - x.first + 1;
      for (uint64_t* f : call_fields(once)) *f *= n;
      add_call_stats(s, once);
      s.instance_variables += ins * n;
      s.witness_variables += wit * n;
      break;
    }
    default: break;
  }
}

std::string Stats::to_json_pretty() const {
  std::string out = "{\n  \"field_characteristic\": ";
  if (field_characteristic.empty()) {
    out += "[]";
  } else {
    out += "[\n";
    for (size_t i = 0; i < field_characteristic.size(); ++i) {
      out += "    " + std::to_string((unsigned)field_characteristic[i]);
      out += i + 1 < field_characteristic.size() ? ",\n" : "\n";
    }
    out += "  ]";
  }
  out += ",\n  \"field_degree\": " + std::to_string(field_degree) + ",\n  \"gate_stats\": ";
  write_gate_stats(out, gate_stats, "  ");
  out += ",\n  \"functions\": ";
  if (functions.empty()) {
    out += "{}";
  } else {
    out += "{\n";
    size_t k = 0;
    for (const auto& kv : functions) {
      out += "    " + json_string(kv.first) + ": [\n      ";
      write_gate_stats(out, kv.second.stats, "      ");
      out += ",\n      " + std::to_string(kv.second.instance_count) + ",\n      " +
             std::to_string(kv.second.witness_count) + "\n    ]";
      out += ++k < functions.size() ? ",\n" : "\n";
    }
    out += "  }";
  }
  out += "\n}";
  return out;
}

}  // namespace zki
