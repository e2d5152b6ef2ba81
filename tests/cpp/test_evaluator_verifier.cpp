// The Evaluator is backend-agnostic, like the reference's (evaluator.rs:1007-1080
// test_evaluator_as_verifier): it runs with an arbitrary ZKBackend whose wires carry no value, and
// without any witness message.  Built and run by tests/test_c_abi.py with plain g++ (host code only).
#include <stdio.h>
#include <fstream>
#include <iterator>

#include "evaluator.hpp"

using namespace zki;

struct VerifierBackend {  // Wire = i64, every wire 0; FieldElement = the raw bytes
  using Wire = long long;
  using FieldElement = Value;
  int calls = 0, asserts = 0, witnesses_without_value = 0;
  static FieldElement from_bytes_le(const Value& v) { return v; }
  void set_field(const Value&, uint32_t, bool) {}
  FieldElement one() const { return Value{1}; }
  FieldElement minus_one() const { return Value{100}; }
  FieldElement zero() const { return Value{0}; }
  Wire copy(const Wire&) { ++calls; return 0; }
  Wire constant(FieldElement) { ++calls; return 0; }
  void assert_zero(const Wire&) { ++asserts; }
  Wire add(const Wire&, const Wire&) { ++calls; return 0; }
  Wire multiply(const Wire&, const Wire&) { ++calls; return 0; }
  Wire add_constant(const Wire&, FieldElement) { ++calls; return 0; }
  Wire mul_constant(const Wire&, FieldElement) { ++calls; return 0; }
  Wire and_(const Wire&, const Wire&) { ++calls; return 0; }
  Wire xor_(const Wire&, const Wire&) { ++calls; return 0; }
  Wire not_(const Wire&) { ++calls; return 0; }
  Wire instance(FieldElement) { ++calls; return 0; }
  Wire witness(const FieldElement* v) {  // verifier mode: the value is absent
    ++calls;
    if (!v) ++witnesses_without_value;
    return 0;
  }
};

static std::vector<uint8_t> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;  // instance.sieve relation.sieve  (no witness on purpose)
  VerifierBackend backend;
  Evaluator<VerifierBackend> ev;
  for (int k = 1; k <= 2; ++k) {
    std::vector<uint8_t> bytes = slurp(argv[k]);
    for (const auto& m : split_messages(bytes.data(), bytes.size())) ev.ingest_buffer(bytes.data() + m.first, m.second, backend);
  }
  const auto v = ev.get_violations();
  for (const auto& s : v) fprintf(stderr, "violation: %s\n", s.c_str());
  printf("calls %d asserts %d witnesses_without_value %d violations %zu\n", backend.calls, backend.asserts,
         backend.witnesses_without_value, v.size());
  return v.empty() ? 0 : 1;
}
