#include "validator.hpp"

#include <algorithm>
#include <array>

namespace zki {

const char* const kNamesRegexText = "^[a-zA-Z_][\\w]*(?:(?:\\.|:{2})[a-zA-Z_][\\w]*)*$";  // validator.rs:25

namespace {

// UTF-8 -> code points; malformed bytes become U+FFFD (Rust strings are always valid UTF-8).
std::vector<uint32_t> code_points(const std::string& s) {
  std::vector<uint32_t> out;
  for (size_t i = 0; i < s.size();) {
    const uint8_t c = (uint8_t)s[i];
    int extra = c < 0x80 ? 0 : (c >> 5) == 6 ? 1 : (c >> 4) == 14 ? 2 : (c >> 3) == 30 ? 3 : -1;
    if (extra < 0 || i + extra >= s.size() + (extra == 0)) {
      out.push_back(extra == 0 ? c : 0xFFFD);
      ++i;
      continue;
    }
    uint32_t cp = extra == 0 ? c : c & (0x3F >> extra);
    for (int k = 1; k <= extra; ++k) cp = (cp << 6) | ((uint8_t)s[i + k] & 0x3F);
    out.push_back(cp);
    i += extra + 1;
  }
  return out;
}

bool is_digit(uint32_t c) { return c >= '0' && c <= '9'; }
bool is_name_start(uint32_t c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_'; }
bool is_word(uint32_t c) {
  if (c < 0x80) return is_name_start(c) || is_digit(c);
  if (c < 0x100) return c == 0xAA || c == 0xB5 || c == 0xBA || (c >= 0xC0 && c != 0xD7 && c != 0xF7);
  if (c >= 0x2000 && c <= 0x2BFF) return false;  // punctuation, symbols, arrows, operators
  if (c >= 0x3000 && c <= 0x303F) return false;  // CJK punctuation
  if (c >= 0xE000 && c <= 0xF8FF) return false;  // private use
  if (c >= 0xFFF0 && c <= 0xFFFF) return false;
  return true;
}

bool is_space(uint32_t c) {  // char::is_whitespace
  return c == ' ' || (c >= 9 && c <= 13) || c == 0x85 || c == 0xA0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200A) ||
         c == 0x2028 || c == 0x2029 || c == 0x202F || c == 0x205F || c == 0x3000;
}

std::vector<uint32_t> trimmed(const std::string& s) {  // str::trim
  std::vector<uint32_t> cp = code_points(s);
  size_t a = 0, b = cp.size();
  while (a < b && is_space(cp[a])) ++a;
  while (b > a && is_space(cp[b - 1])) --b;
  return std::vector<uint32_t>(cp.begin() + a, cp.begin() + b);
}

bool match_version(const std::vector<uint32_t>& s) {
  // ^\d+.\d+.\d+$ : split the string as D1 x D2 y D3 with every Di a non-empty digit run
  const size_t n = s.size();
  std::vector<size_t> run(n + 1, 0);  // run[i] = length of the digit run starting at i
  for (size_t i = n; i-- > 0;) run[i] = is_digit(s[i]) ? run[i + 1] + 1 : 0;
  for (size_t a = 1; a <= (n ? run[0] : 0); ++a) {        // D1 = s[0..a)
    if (a >= n || s[a] == '\n') continue;                  // x = s[a]
    for (size_t b = 1; a + 1 + b <= n && b <= run[a + 1]; ++b) {  // D2 = s[a+1 .. a+1+b)
      const size_t y = a + 1 + b;
      if (y >= n || s[y] == '\n') continue;
      if (y + 1 < n && run[y + 1] == n - (y + 1)) return true;  // D3 = the rest, all digits
    }
  }
  return false;
}

bool match_name(const std::vector<uint32_t>& s) {
  size_t i = 0;
  const size_t n = s.size();
  auto segment = [&]() {
    if (i >= n || !is_name_start(s[i])) return false;
    ++i;
    while (i < n && is_word(s[i])) ++i;
    return true;
  };
  if (!segment()) return false;
  while (i < n) {
    if (s[i] == '.') ++i;
    else if (s[i] == ':' && i + 1 < n && s[i + 1] == ':') i += 2;
    else return false;
    if (!segment()) return false;
  }
  return true;
}

std::string debug_bytes(const Value& v) {  // {:?} of a Vec<u8>
  std::string s = "[";
  for (size_t i = 0; i < v.size(); ++i) {
    if (i) s += ", ";
    s += std::to_string((unsigned)v[i]);
  }
  return s + "]";
}

}  // namespace

bool matches_version_pattern(const std::string& s) { return match_version(code_points(s)); }
bool matches_name_pattern(const std::string& s) { return match_name(code_points(s)); }

Validator::Validator()
    : known_functions_(std::make_shared<FunctionTable>()),
      known_iterators_(std::make_shared<IteratorScope>()),
      steps_left_(std::make_shared<uint64_t>(1ull << 40)) {}

void Validator::step(uint64_t n) {
  if (*steps_left_ < n) throw Error("Validator: step limit exceeded");
  *steps_left_ -= n;
}

std::vector<std::string> Validator::get_violations() const {
  std::vector<std::string> out = violations_;
  if (instance_queue_len_ > 0)  // ensure_all_instance_values_consumed (:839-846)
    out.push_back("Too many Instance values (" + std::to_string(instance_queue_len_) + " not consumed)");
  if (as_prover_ && witness_queue_len_ > 0)  // :848-855
    out.push_back("Too many Witness values (" + std::to_string(witness_queue_len_) + " not consumed)");
  return out;
}

void Validator::ingest_message(const Message& msg) {
  switch (msg.kind) {
    case Message::IsInstance: ingest_instance(msg.instance); break;
    case Message::IsWitness: ingest_witness(msg.witness); break;
    case Message::IsRelation: ingest_relation(msg.relation); break;
    default: break;
  }
}

void Validator::ingest_header(const Header& header) {
  if (got_header_) {
    if (field_characteristic_ != BigNat::from_bytes_le(header.field_characteristic))
      violate("The field_characteristic field is not consistent across headers.");
    if (field_degree_ != header.field_degree) violate("The field_degree is not consistent across headers.");
    if (header_version_ != header.version) violate("The profile version is not consistent across headers.");
    return;
  }
  got_header_ = true;
  field_characteristic_ = BigNat::from_bytes_le(header.field_characteristic);
  if (!(field_characteristic_.cmp(BigNat(1)) > 0)) violate("The field_characteristic should be > 1");
  if (!is_probably_prime(header.field_characteristic)) violate("The field_characteristic should be a prime.");
  field_degree_ = header.field_degree;
  if (field_degree_ != 1) violate("field_degree must be = 1");
  if (!match_version(trimmed(header.version)))
    violate("The profile version should match the following format <major>.<minor>.<patch>.");
  header_version_ = header.version;
}

void Validator::ingest_instance(const Instance& instance) {
  ingest_header(instance.header);
  for (const Value& v : instance.common_inputs) ensure_value_in_field(v, "instance value " + debug_bytes(v));
  instance_queue_len_ += instance.common_inputs.size();
}

void Validator::ingest_witness(const Witness& witness) {
  if (!as_prover_) violate("As verifier, got an unexpected Witness message.");
  ingest_header(witness.header);
  for (const Value& v : witness.short_witness) ensure_value_in_field(v, "witness value " + debug_bytes(v));
  witness_queue_len_ += witness.short_witness.size();
}

void Validator::ingest_relation(const Relation& relation) {
  ingest_header(relation.header);

  gate_set_ = relation.gate_mask;
  if (mask::contains_feature(gate_set_, mask::BOOL) && mask::contains_feature(gate_set_, mask::ARITH))
    violate("Cannot mix arithmetic and boolean gates");
  if (mask::contains_feature(gate_set_, mask::BOOL) && field_characteristic_ != BigNat(2))
    violate("With boolean profile the field characteristic can only be 2.");
  features_ = relation.feat_mask;

  for (const Function& f : relation.functions) {
    ensure_allowed_feature("@function", mask::FUNCTION);
    if (!match_name(trimmed(f.name)))
      violate("The function name (" + f.name + ") should match the proper format (" + kNamesRegexText + ").");
    if (known_functions_->count(f.name)) {
      violate("A function with the name '" + f.name + "' already exists");
      continue;
    }
    (*known_functions_)[f.name] = {f.output_count, f.input_count, f.instance_count, f.witness_count};
    static const Subcircuit kEmpty;
    ingest_subcircuit(f.body ? *f.body : kEmpty, f.output_count, f.input_count, f.instance_count, f.witness_count, false);
  }
  for (const Gate& g : relation.gates) ingest_gate(g);
}

std::vector<WireId> Validator::expand_or_violate(const WireList& l) {
  try {
    std::vector<WireId> v = expand_wirelist(l);
    step(v.size());
    return v;
  } catch (const Error& e) {
    violate(e.what());
    return {};
  }
}

void Validator::ingest_gate(const Gate& g) {
  step();
  static const Subcircuit kEmpty;
  switch (g.kind) {
    case GateKind::Constant:
      ensure_value_in_field(g.ext ? g.ext->constant : Value(), "Gate::Constant constant");
      ensure_undefined_and_set(g.out);
      break;
    case GateKind::AssertZero:
      ensure_defined_and_set(g.in0);
      break;
    case GateKind::Copy:
      ensure_defined_and_set(g.in0);
      ensure_undefined_and_set(g.out);
      break;
    case GateKind::Add:
    case GateKind::Mul:
    case GateKind::And:
    case GateKind::Xor: {
      const char* name = g.kind == GateKind::Add ? "@add" : g.kind == GateKind::Mul ? "@mul" : g.kind == GateKind::And ? "@and" : "@xor";
      const uint16_t bit = g.kind == GateKind::Add ? mask::ADD : g.kind == GateKind::Mul ? mask::MUL : g.kind == GateKind::And ? mask::AND : mask::XOR;
      ensure_allowed_gate(name, bit);
      ensure_defined_and_set(g.in0);
      ensure_defined_and_set(g.in1);
      ensure_undefined_and_set(g.out);
      break;
    }
    case GateKind::AddConstant:
    case GateKind::MulConstant: {
      const bool add = g.kind == GateKind::AddConstant;
      ensure_allowed_gate(add ? "@addc" : "@mulc", add ? mask::ADDC : mask::MULC);
      ensure_value_in_field(g.ext ? g.ext->constant : Value(),
                            std::string(add ? "Gate::AddConstant_" : "Gate::MulConstant_") + std::to_string(g.out));
      ensure_defined_and_set(g.in0);
      ensure_undefined_and_set(g.out);
      break;
    }
    case GateKind::Not:
      ensure_allowed_gate("@not", mask::NOT);
      ensure_defined_and_set(g.in0);
      ensure_undefined_and_set(g.out);
      break;
    case GateKind::Instance:
      declare(g.out);
      consume_instance(1);
      break;
    case GateKind::Witness:
      declare(g.out);
      consume_witness(1);
      break;
    case GateKind::Free: {
      const WireId first = g.in0, last = g.has_last ? g.in1 : g.in0;
      if (g.has_last && last <= first)
        violate("For Free gates, last WireId (" + std::to_string(last) + ") must be strictly greater than first WireId (" +
                std::to_string(first) + ").");
      for (WireId w = first; w <= last; ++w) {
        step();
        ensure_defined_and_set(w);
        remove(w);
        if (w == UINT64_MAX) break;
      }
      break;
    }
    case GateKind::AnonCall: {
      const GateExt& x = *g.ext;
      ensure_allowed_feature("@anoncall", mask::FUNCTION);
      const std::vector<WireId> outs = expand_or_violate(x.output_wires), ins = expand_or_violate(x.input_wires);
      for (WireId id : ins) ensure_defined_and_set(id);
      ingest_subcircuit(x.subcircuit ? *x.subcircuit : kEmpty, outs.size(), ins.size(), x.instance_count, x.witness_count, true);
      consume_instance(x.instance_count);
      consume_witness(x.witness_count);
      for (WireId id : outs) ensure_undefined_and_set(id);
      break;
    }
    case GateKind::Call: {
      const GateExt& x = *g.ext;
      ensure_allowed_feature("@call", mask::FUNCTION);
      const std::vector<WireId> outs = expand_or_violate(x.output_wires), ins = expand_or_violate(x.input_wires);
      for (WireId id : ins) ensure_defined_and_set(id);
      uint64_t ic = 0, wc = 0;
      if (!ingest_call(x.name, outs.size(), ins.size(), &ic, &wc)) ic = wc = 0;
      consume_instance(ic);
      consume_witness(wc);
      for (WireId id : outs) ensure_undefined_and_set(id);
      break;
    }
    case GateKind::Switch: {
      const GateExt& x = *g.ext;
      ensure_allowed_feature("@switch", mask::SWITCH);
      ensure_defined_and_set(g.in0);
      if (x.cases.size() != x.branches.size())
        violate("Gate::Switch: The number of cases value does not match the number of branches.");
      if (x.cases.empty()) {
        if (!x.output_wires.empty()) violate("Switch: no case given while non-empty list of output wires.");
        return;
      }
      std::vector<BigNat> seen;
      for (const Value& c : x.cases) {
        const BigNat v = BigNat::from_bytes_le(c);
        ensure_value_in_field(c, "Gate::Switch case value: " + v.to_decimal());
        if (std::find(seen.begin(), seen.end(), v) == seen.end()) seen.push_back(v);
      }
      if (seen.size() != x.cases.size()) violate("Gate::Switch: The cases values contain duplicates.");

      uint64_t max_ic = 0, max_wc = 0;
      const std::vector<WireId> outs = expand_or_violate(x.output_wires);
      for (const CaseInvoke& br : x.branches) {
        uint64_t ic = 0, wc = 0;
        const std::vector<WireId> ins = expand_or_violate(br.input_wires);
        for (WireId id : ins) ensure_defined_and_set(id);
        if (!br.anonymous) {
          if (!ingest_call(br.name, outs.size(), ins.size(), &ic, &wc)) ic = wc = 0;
        } else {
          ingest_subcircuit(br.subcircuit ? *br.subcircuit : kEmpty, outs.size(), ins.size(), br.instance_count,
                            br.witness_count, true);
          ic = br.instance_count;
          wc = br.witness_count;
        }
        max_ic = std::max(max_ic, ic);
        max_wc = std::max(max_wc, wc);
      }
      consume_instance(max_ic);
      consume_witness(max_wc);
      for (WireId id : outs) ensure_undefined_and_set(id);
      break;
    }
    case GateKind::For: {
      const GateExt& x = *g.ext;
      ensure_allowed_feature("@for", mask::FOR);
      if (x.last < x.first) {
        violate("In a For loop, the end value (" + std::to_string(x.last) +
                ") must be strictly greater than the start value (" + std::to_string(x.first) + ").");
        return;
      }
      if (known_iterators_->find(x.name)) {
        violate("Iterator already used in this context.");
        return;
      }
      if (!matches_name_pattern(x.name))
        violate("The iterator name (" + x.name + ") should match the following format (" + kNamesRegexText + ").");
      for (uint64_t i = x.first; i <= x.last; ++i) {
        step();
        known_iterators_->insert(x.name, i);
        const ForLoopBody& body = x.body;
        const std::vector<WireId> outs = evaluate_iterexpr_list(body.outputs, *known_iterators_);
        const std::vector<WireId> ins = evaluate_iterexpr_list(body.inputs, *known_iterators_);
        step(outs.size() + ins.size());
        for (WireId id : ins) ensure_defined_and_set(id);
        if (!body.anonymous) {
          uint64_t ic = 0, wc = 0;
          if (!ingest_call(body.name, outs.size(), ins.size(), &ic, &wc)) ic = wc = 0;
          for (WireId id : outs) ensure_undefined_and_set(id);
          consume_instance(ic);
          consume_witness(wc);
        } else {
          ingest_subcircuit(body.subcircuit ? *body.subcircuit : kEmpty, outs.size(), ins.size(), body.instance_count,
                            body.witness_count, true);
          for (WireId id : outs) ensure_undefined_and_set(id);
          consume_instance(body.instance_count);
          consume_witness(body.witness_count);
        }
        if (i == UINT64_MAX) break;
      }
      known_iterators_->remove(x.name);
      for (WireId id : expand_or_violate(x.output_wires)) ensure_defined_and_set(id);
      break;
    }
    default:
      break;
  }
}

bool Validator::ingest_call(const std::string& name, size_t n_out, size_t n_in, uint64_t* ins, uint64_t* wit) {
  const auto it = known_functions_->find(name);
  if (it == known_functions_->end()) {
    violate("Unknown Function gate " + name);
    return false;
  }
  if (it->second[0] != n_out) violate("Call: number of output wires mismatch.");
  if (it->second[1] != n_in) violate("Call: number of input wires mismatch.");
  *ins = it->second[2];
  *wit = it->second[3];
  return true;
}

void Validator::ingest_subcircuit(const Subcircuit& sub, uint64_t output_count, uint64_t input_count,
                                  uint64_t instance_count, uint64_t witness_count, bool use_same_scope) {
  Validator inner;
  inner.as_prover_ = as_prover_;
  inner.instance_queue_len_ = instance_count;
  inner.witness_queue_len_ = as_prover_ ? witness_count : 0;
  inner.got_header_ = got_header_;
  inner.gate_set_ = gate_set_;
  inner.features_ = features_;
  inner.header_version_ = header_version_;
  inner.field_characteristic_ = field_characteristic_;
  inner.field_degree_ = field_degree_;
  inner.known_functions_ = known_functions_;
  if (use_same_scope) inner.known_iterators_ = known_iterators_;
  inner.steps_left_ = steps_left_;

  // inputs are numbered from output_count on and are defined on entry
  step(input_count);
  for (uint64_t w = output_count; w < output_count + input_count; ++w) inner.live_wires_.insert(w);
  for (const Gate& g : sub) inner.ingest_gate(g);
  step(output_count);
  for (uint64_t w = 0; w < output_count; ++w) inner.ensure_defined_and_set(w);

  violations_.insert(violations_.end(), inner.violations_.begin(), inner.violations_.end());
  if (inner.instance_queue_len_ != 0)
    violate("The subcircuit has not consumed all the instance variables it should have.");
  if (inner.witness_queue_len_ != 0)
    violate("The subcircuit has not consumed all the witness variables it should have.");
}

void Validator::remove(WireId id) {
  if (!live_wires_.erase(id))
    violate("The variable " + std::to_string(id) +
            " is being freed, but was not defined previously, or has been already freed");
}

void Validator::consume_instance(uint64_t n) {
  if (instance_queue_len_ >= n) {
    instance_queue_len_ -= n;
  } else {
    instance_queue_len_ = 0;
    violate("Not enough Instance value to consume.");
  }
}

void Validator::consume_witness(uint64_t n) {
  if (!as_prover_) return;
  if (witness_queue_len_ >= n) {
    witness_queue_len_ -= n;
  } else {
    witness_queue_len_ = 0;
    violate("Not enough Witness value to consume.");
  }
}

void Validator::ensure_defined_and_set(WireId id) {
  if (is_defined(id)) return;
  if (as_prover_)
    violate("The wire " + std::to_string(id) + " is used but was not assigned a value, or has been freed already.");
  declare(id);  // avoids repeating the message for the same wire
}

void Validator::ensure_undefined(WireId id) {
  if (is_defined(id))
    violate("The wire " + std::to_string(id) + " has already been initialized before. This violates the SSA property.");
}

void Validator::ensure_undefined_and_set(WireId id) {
  ensure_undefined(id);
  declare(id);
}

void Validator::ensure_value_in_field(const Value& value, const std::string& name) {
  if (value.empty()) violate("The " + name + " is empty.");
  const BigNat v = BigNat::from_bytes_le(value);
  if (v >= field_characteristic_)
    violate("The " + name + " cannot be represented in the field specified in Header (" + v.to_decimal() +
            " >= " + field_characteristic_.to_decimal() + ").");
}

void Validator::ensure_allowed_gate(const char* name, uint16_t mask_bit) {
  if (!mask::contains_feature(gate_set_, mask_bit))
    violate(std::string("The gate ") + name + " is not allowed in this circuit.");
}

void Validator::ensure_allowed_feature(const char* name, uint16_t mask_bit) {
  if (!mask::contains_feature(features_, mask_bit))
    violate(std::string("The feature ") + name + " is not allowed in this circuit.");
}

const char* Validator::implemented_checks() {
  return "Header: characteristic > 1 and prime, degree 1, version <major>.<minor>.<patch>, headers coherent.\n"
         "Relation: gateset arithmetic or boolean (boolean only over characteristic 2).\n"
         "Inputs: enough Instance / Witness values, all consumed, every value below the characteristic.\n"
         "Gates: allowed by the gateset, constants in the field, inputs defined, single static assignment,\n"
         "       @function/@for/@switch only when enabled, Free / For / WireRange bounds ordered.\n";
}

}  // namespace zki
