// Sanitizer driver for the host side of the product (CPU build only: GPU AddressSanitizer is not
// available on the pool).  Built by tests/test_sanitizers.py with -fsanitize=address,undefined from the
// product's own sources (reader, Evaluator, TapeBackend, Validator, Stats, scheduler, R1CS emission) and fed
// statements from files: every file is a stream of size-prefixed messages.  Errors the code reports are fine;
// the process must not trip a sanitizer.
//   fuzz_host <max_tape_ops> <file>...
#include <stdio.h>
#include <stdlib.h>

#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "evaluator.hpp"
#include "r1cs.hpp"
#include "schedule.hpp"
#include "sieve/reader.hpp"
#include "stats.hpp"
#include "tape.hpp"
#include "validator.hpp"

using namespace zki;

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const uint64_t max_ops = strtoull(argv[1], nullptr, 10);
  size_t n_errors = 0, n_scheduled = 0;
  for (int k = 2; k < argc; ++k) {
    std::ifstream f(argv[k], std::ios::binary);
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    TapeBackend backend;
    backend.set_max_ops(max_ops);
    Evaluator<TapeBackend> ev;
    Validator validator = Validator::new_as_prover();
    validator.set_max_steps(max_ops);
    Stats stats;
    try {
      for (const auto& m : split_messages(data.data(), data.size())) {
        Message msg;
        try {
          msg = read_message(data.data() + m.first, m.second);
        } catch (const std::exception&) {
          ++n_errors;
          continue;
        }
        try {
          validator.ingest_message(msg);
        } catch (const std::exception&) {
          ++n_errors;
        }
        stats.ingest_message(msg);
        if (msg.kind == Message::IsInstance) {
          ev.set_modulus(msg.instance.header.field_characteristic);
          for (const Value& v : msg.instance.common_inputs) ev.push_instance(backend.import_instance(v));
        } else if (msg.kind == Message::IsWitness) {
          ev.set_modulus(msg.witness.header.field_characteristic);
          for (const Value& v : msg.witness.short_witness) ev.push_witness(backend.import_witness(v));
        } else {
          ev.ingest_message(msg, backend);
        }
      }
    } catch (const std::exception&) {
      ++n_errors;
    }
    n_errors += ev.get_violations().size() + validator.get_violations().size();
    (void)stats.to_json_pretty();
    if (backend.field_set() && backend.tape().size() != 0) {
      for (int variant = 0; variant < 3; ++variant) {
        try {
          ScheduleOptions opt;
          opt.retain_all = variant == 0;
          opt.sort_by_operand = variant;
          Schedule s = build_schedule(backend.tape(), backend.field(), opt);
          n_scheduled += s.n_levels != 0;
        } catch (const std::exception&) {
          ++n_errors;
        }
      }
      try {
        Value mod;
        for (uint32_t i = 0; i < backend.field().nwords; ++i)
          for (int b = 0; b < 4; ++b) mod.push_back((uint8_t)(backend.field().p[i] >> (8 * b)));
        R1cs r = r1cs_from_tape(backend.tape(), backend.field(), mod, false);
        (void)r;
      } catch (const std::exception&) {
        ++n_errors;
      }
    }
  }
  printf("files=%d errors=%zu scheduled=%zu\n", argc - 2, n_errors, n_scheduled);
  return 0;
}
