#include "r1cs.hpp"

#include <map>

namespace zki {
namespace {

Value strip(const Value& v) {  // set_field trims trailing zeros of the modulus (to_r1cs.rs:104-106)
  Value r = v;
  while (!r.empty() && r.back() == 0) r.pop_back();
  return r;
}
// BigUint::to_bytes_le(): minimal little-endian bytes, [0] for zero
Value minimal(const Value& v) {
  Value r = strip(v);
  if (r.empty()) r.push_back(0);
  return r;
}

}  // namespace

R1cs r1cs_from_tape(const Tape& tape, const FieldHost& field, const Value& modulus, bool use_correction) {
  (void)field;
  R1cs out;
  std::map<Value, uint32_t> interned;
  auto coef = [&](const Value& bytes) {
    auto it = interned.find(bytes);
    if (it != interned.end()) return it->second;
    const uint32_t idx = (uint32_t)out.coefs.size();
    out.coefs.push_back(bytes);
    interned.emplace(bytes, idx);
    return idx;
  };
  const uint32_t c_one = coef(Value{1});
  const uint32_t c_zero = coef(Value{0});
  const uint32_t c_mod = coef(strip(modulus));
  out.var_kind.push_back(0);
  auto new_var = [&](uint8_t kind) {
    out.var_kind.push_back(kind);
    return out.n_vars++;
  };
  auto begin_row = [&] { out.row_ptr.push_back((uint32_t)out.terms.size()); };
  auto term = [&](uint64_t var, uint32_t c) { out.terms.push_back(R1csTerm{var, c}); };
  out.var_of_op.assign(tape.size(), kNoVar);
  for (size_t i = 0; i < tape.size(); ++i) {
    const uint8_t k = tape.kind[i];
    switch (k) {
      case TK_COPY: out.var_of_op[i] = out.var_of_op[tape.a[i]]; break;                    // :143-145
      case TK_CONST: out.var_of_op[i] = new_var(1); break;                                  // :147-153 instance var
      case TK_INSTANCE: out.var_of_op[i] = new_var(1); break;                               // :373-379
      case TK_WITNESS: out.var_of_op[i] = new_var(2); break;                                // :381-392
      case TK_ASSERT:                                                                        // :155-161
        begin_row(); term(out.var_of_op[tape.a[i]], c_one);
        begin_row(); term(kVarOne, c_one);
        begin_row(); term(kVarOne, c_zero);
        break;
      case TK_ADD: case TK_XOR: case TK_ADDC: case TK_NOT: case TK_MUL: case TK_AND: case TK_MULC: {
        const uint64_t res = new_var(3);
        const uint64_t corr = use_correction ? new_var(4) : 0;
        out.var_of_op[i] = res;
        const uint64_t a = out.var_of_op[tape.a[i]];
        const bool additive = k == TK_ADD || k == TK_XOR || k == TK_ADDC || k == TK_NOT;
        auto out_terms = [&] {  // [out*1 (+ correction*modulus)]
          term(res, c_one);
          if (use_correction) term(corr, c_mod);
        };
        if (additive) {  // :163-211, :262-312 ; xor = add (:365-367), not = add_constant(a, 1) (:369-371)
          begin_row(); out_terms();
          begin_row(); term(kVarOne, c_one);
          begin_row(); term(a, c_one);
          if (k == TK_ADD || k == TK_XOR) term(out.var_of_op[tape.b[i]], c_one);
          else if (k == TK_NOT) term(kVarOne, c_one);
          else term(kVarOne, coef(minimal(tape.consts[tape.b[i]])));
        } else if (k == TK_MULC) {  // :314-359
          begin_row(); term(a, coef(minimal(tape.consts[tape.b[i]])));
          begin_row(); term(kVarOne, c_one);
          begin_row(); out_terms();
        } else {  // multiply / and (:213-260, :361-363)
          begin_row(); term(a, c_one);
          begin_row(); term(out.var_of_op[tape.b[i]], c_one);
          begin_row(); out_terms();
        }
        break;
      }
      default: break;
    }
  }
  out.row_ptr.push_back((uint32_t)out.terms.size());
  return out;
}

}  // namespace zki
