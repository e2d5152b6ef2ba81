#include "schedule.hpp"

#include "sieve/bignum.hpp"

#include <string.h>

#include <algorithm>

namespace zki {
namespace {

inline int n_inputs(uint8_t k) {
  switch (k) {
    case TK_ADD: case TK_MUL: case TK_AND: case TK_XOR: return 2;
    case TK_ADDC: case TK_MULC: case TK_COPY: case TK_NOT: case TK_ASSERT: case TK_NZ: return 1;
    default: return 0;  // CONST, INSTANCE, WITNESS, NOP
  }
}

constexpr uint32_t kInf = 0xFFFFFFFFu;

}  // namespace

namespace {

// Depth-first walk of the "reads the same wire" graph of one level: ops become neighbours of an op they
// share an operand with, so the second reader of a wire runs while the first reader's fetch is still in
// the XCD's L2 (device/replay_kernels.hpp block_coords keeps the ops of a lane block on one XCD).  On the
// random wiring of the C2 workload this takes the distinct-line fetches of a level from 1.6 to 1.07 per op.
template <class Gathered>
void locality_order(uint32_t* ops, size_t cnt, Gathered&& gathered) {
  struct Edge {
    uint32_t wire, pos;
  };
  std::vector<Edge> edges;
  edges.reserve(cnt * 2);
  std::vector<uint32_t> first_edge(cnt + 1, 0);
  for (size_t p = 0; p < cnt; ++p) {
    uint32_t w[4];
    const int k = gathered(ops[p], w);
    for (int j = 0; j < k; ++j) edges.push_back({w[j], (uint32_t)p});
  }
  std::sort(edges.begin(), edges.end(), [](const Edge& x, const Edge& y) { return x.wire != y.wire ? x.wire < y.wire : x.pos < y.pos; });
  // wire rank -> [begin, end) in `edges`; per op the ranks of its wires
  std::vector<uint32_t> wire_begin;
  std::vector<uint32_t> op_wires(cnt * 4, 0xFFFFFFFFu);
  std::vector<uint8_t> op_nw(cnt, 0);
  for (size_t e = 0; e < edges.size(); ++e) {
    if (e == 0 || edges[e].wire != edges[e - 1].wire) wire_begin.push_back((uint32_t)e);
    const uint32_t rank = (uint32_t)wire_begin.size() - 1, p = edges[e].pos;
    if (op_nw[p] < 4 && (op_nw[p] == 0 || op_wires[4 * p + op_nw[p] - 1] != rank)) op_wires[4 * p + op_nw[p]++] = rank;
  }
  wire_begin.push_back((uint32_t)edges.size());
  std::vector<uint8_t> visited(cnt, 0), expanded(wire_begin.size(), 0);
  std::vector<uint32_t> out, stack;
  out.reserve(cnt);
  for (size_t root = 0; root < cnt; ++root) {
    if (visited[root]) continue;
    stack.push_back((uint32_t)root);
    while (!stack.empty()) {
      const uint32_t p = stack.back();
      stack.pop_back();
      if (visited[p]) continue;
      visited[p] = 1;
      out.push_back(ops[p]);
      for (int j = op_nw[p] - 1; j >= 0; --j) {
        const uint32_t r = op_wires[4 * p + j];
        const uint32_t b = wire_begin[r], e = wire_begin[r + 1];
        // a wire with many readers is expanded once; re-pushing its readers at every visit would be quadratic
        if (e - b > 8) {
          if (expanded[r]) continue;
          expanded[r] = 1;
        }
        for (uint32_t q = e; q-- > b;)
          if (!visited[edges[q].pos]) stack.push_back(edges[q].pos);
      }
    }
  }
  for (size_t p = 0; p < cnt; ++p) ops[p] = out[p];
}

}  // namespace

namespace {

// Switch weights are 1 - (case - cond)^(p-1) (evaluator.rs:823-839); the reference computes the power with a
// square-and-multiply ladder (bits(p) squarings + popcount(p-1) multiplies: 352 dependent products at BN254, each of
// them a level of its own).  For a prime p Fermat gives x^(p-1) = 1 for x != 0 and 0 for x = 0, so in the production
// schedule the ladder's result becomes one entry `x != 0` and the ladder's own ops are dropped.  Only done when the
// characteristic passes the primality test (sieve/bignum.cpp) and nothing outside a ladder reads its intermediates.
bool rewrite_ladders(const Tape& in, const FieldHost& field, Tape* out, uint64_t* n_done) {
  if (in.ladders.empty()) return false;
  Value p_le;
  for (uint32_t i = 0; i < field.nwords; ++i)
    for (int b = 0; b < 4; ++b) p_le.push_back((uint8_t)(field.p[i] >> (8 * b)));
  if (!is_probably_prime(p_le)) return false;
  const size_t n = in.size();
  // bits of the exponent p - 1 (p is odd: clear bit 0)
  uint32_t e[kFieldWords];
  for (int i = 0; i < kFieldWords; ++i) e[i] = i < (int)field.nwords ? field.p[i] : 0;
  e[0] &= ~1u;
  int top_bit = -1;
  for (int i = 32 * kFieldWords - 1; i >= 0 && top_bit < 0; --i)
    if ((e[i / 32] >> (i % 32)) & 1) top_bit = i;
  if (top_bit < 0) return false;
  // A hint is only a hint (zkgpu_backend_ladder is a public entry): the range must BE the reference's
  // square-and-multiply recursion over p - 1 (evaluator.rs:801-820) -- copy(base), then from the second highest bit
  // down a squaring of the running value and, where the bit is set, a multiply by the base -- or it stays as recorded.
  auto is_ladder = [&](const Tape::Ladder& L) {
    if (L.result >= n || L.first > L.result || L.base >= L.first) return false;
    uint32_t cur = L.first;
    if (in.kind[cur] != TK_COPY || in.a[cur] != L.base) return false;
    for (int shift = top_bit - 1; shift >= 0; --shift) {
      if (++cur > L.result || in.kind[cur] != TK_MUL || in.a[cur] != cur - 1 || in.b[cur] != cur - 1) return false;
      if ((e[shift / 32] >> (shift % 32)) & 1) {
        if (++cur > L.result || in.kind[cur] != TK_MUL) return false;
        const bool ab = in.a[cur] == cur - 1 && in.b[cur] == L.base, ba = in.a[cur] == L.base && in.b[cur] == cur - 1;
        if (!ab && !ba) return false;
      }
    }
    return cur == L.result;
  };
  std::vector<uint32_t> owner(n, kInf);  // ladder whose intermediate this op is
  std::vector<uint8_t> ok(in.ladders.size(), 1);
  for (size_t l = 0; l < in.ladders.size(); ++l) {
    const Tape::Ladder& L = in.ladders[l];
    if (!is_ladder(L)) { ok[l] = 0; continue; }
    for (uint32_t i = L.first; i <= L.result; ++i) {
      if (owner[i] != kInf) {  // two hints over one op: neither is trusted, the first keeps the op
        ok[l] = 0;
        ok[owner[i]] = 0;
      } else {
        owner[i] = (uint32_t)l;
      }
    }
  }
  // the result is the one value others may read
  for (size_t l = 0; l < in.ladders.size(); ++l)
    if (in.ladders[l].result < n && owner[in.ladders[l].result] == l) owner[in.ladders[l].result] = kInf;
  for (size_t j = 0; j < n; ++j) {
    const int ni = n_inputs(in.kind[j]);
    for (int k = 0; k < ni; ++k) {
      const uint32_t src = k == 0 ? in.a[j] : in.b[j];
      if (src >= n || owner[src] == kInf) continue;
      const Tape::Ladder& L = in.ladders[owner[src]];
      if (j < L.first || j > L.result) ok[owner[src]] = 0;  // read from outside the ladder
    }
  }
  bool any = false;
  for (size_t l = 0; l < in.ladders.size(); ++l) any = any || ok[l];
  if (!any) return false;
  out->kind = in.kind;
  out->a = in.a;
  out->b = in.b;
  out->assert_op = in.assert_op;
  out->assert_wire = in.assert_wire;
  out->consts = in.consts;
  out->n_instance = in.n_instance;
  out->n_witness = in.n_witness;
  out->n_value_ops = in.n_value_ops;
  for (size_t l = 0; l < in.ladders.size(); ++l) {
    if (!ok[l]) continue;
    const Tape::Ladder& L = in.ladders[l];
    for (uint32_t i = L.first; i < L.result; ++i) out->kind[i] = TK_NOP;
    out->kind[L.result] = TK_NZ;
    out->a[L.result] = L.base;
    out->b[L.result] = 0;
    ++*n_done;
  }
  return true;
}

}  // namespace

namespace {

// The stages of build_schedule(): one object holds the working arrays, one method per stage, run in this order.
struct ScheduleBuilder {
  const Tape& tape;
  const FieldHost& field;
  const ScheduleOptions& opt;
  Schedule s;
  size_t n = 0;
  std::vector<uint8_t> const_odd;       // GF(2): parity of every constant
  uint32_t bool_zero_const = 0;         // GF(2): index of the synthetic constant 0
  std::vector<uint32_t> opa, opb;       // operands resolved through copy chains
  std::vector<uint8_t> absorbed;        // 0 = own entry; 1 = evaluated inside its reader (fusion); 2 = elided (copy, dropped
                                        // ladder op); 3 = shared producer of a pair entry; 4 = second value of a pair entry
  std::vector<uint32_t> first_use, last_use;
  uint32_t n_levels = 0;
  std::vector<uint32_t> pair_second;    // first gate of a pair entry -> second gate (allocated when pairing runs)
  std::vector<uint32_t> order;          // entries in program order
  std::vector<uint64_t> level_start;    // level l = order[level_start[l] .. level_start[l + 1])
  size_t n_live = 0;

  ScheduleBuilder(const Tape& t, const FieldHost& f, const ScheduleOptions& o) : tape(t), field(f), opt(o) {}

  void device_constants();
  void propagate_copies();
  void levelise();
  void fuse_and_pair();
  void order_by_level();
  int gathered(uint32_t i, uint32_t out[4]) const;
  void assign_slots();
  void emit_entries();
  void emit_launches();
};

void ScheduleBuilder::device_constants() {
  // ---- constant pool in device form -------------------------------------
  const uint32_t n_consts = (uint32_t)tape.consts.size();
  const_odd.assign(n_consts, 0);
  bool_zero_const = 0;
  if (s.boolean_path) {
    s.words_per_const = 1;
    s.const_words.resize(n_consts + 1);
    for (uint32_t i = 0; i < n_consts; ++i) {
      const Value& v = tape.consts[i];
      const_odd[i] = !v.empty() && (v[0] & 1);  // value mod 2
      s.const_words[i] = const_odd[i];
    }
    bool_zero_const = n_consts;  // synthetic 0 for mul_constant by an even constant
    s.const_words[n_consts] = 0;
  } else {
    s.words_per_const = field.nwords;
    s.const_words.assign((size_t)n_consts * field.nwords, 0);
    for (uint32_t i = 0; i < n_consts; ++i) {
      uint32_t r[kFieldWords], m[kFieldWords];
      field.reduce(tape.consts[i], r);
      field.to_mont(r, m);
      memcpy(&s.const_words[(size_t)i * field.nwords], m, 4 * field.nwords);
    }
  }
}

void ScheduleBuilder::propagate_copies() {
  // ---- copy propagation ------------------------------------------------------
  // The reference's scoping (ingest_subcircuit, evaluator.rs:698-746) copies every input into and every
  // output out of a call / loop body / switch branch: about half of the backend calls of a structured
  // relation are copies.  A copy has the value of its source, so readers are pointed at the source and a
  // copy nobody can observe any more is not materialised.  Copies that must stay
  // readable (retain_all dumps, wires alive at the end) are kept.
  opa = tape.a;
  opb = tape.b;
  absorbed.assign(n, 0);
  for (size_t i = 0; i < n; ++i)
    if (tape.kind[i] == TK_NOP) absorbed[i] = 2;  // dropped ladder ops: no entry, no slot, read nothing
  const bool propagate = opt.propagate_copies && !opt.retain_all;
  if (propagate) {
    std::vector<uint8_t> is_pinned(n, 0);
    for (uint32_t h : opt.pinned)
      if (h < n) is_pinned[h] = 1;
    std::vector<uint32_t> root(n);
    for (size_t i = 0; i < n; ++i) {
      root[i] = (uint32_t)i;
      if (tape.kind[i] == TK_COPY) root[i] = root[tape.a[i]];  // source of the whole copy chain
    }
    for (size_t i = 0; i < n; ++i) {
      const int ni = n_inputs(tape.kind[i]);
      if (ni >= 1) opa[i] = root[tape.a[i]];
      if (ni == 2) opb[i] = root[tape.b[i]];
      if (tape.kind[i] == TK_COPY && !is_pinned[i]) {
        absorbed[i] = 2;
        ++s.n_copies_elided;
      }
    }
  }
}

void ScheduleBuilder::levelise() {
  // ---- dependency levels (ASAP for ops with inputs) ----------------------
  std::vector<uint32_t>& level = s.level_of;
  first_use.assign(n, kInf);
  last_use.assign(n, 0);
  std::vector<uint8_t> used(n, 0);
  for (size_t i = 0; i < n; ++i) {
    const int ni = n_inputs(tape.kind[i]);
    uint32_t lv = 0;
    if (ni >= 1) lv = level[opa[i]] + 1;
    if (ni == 2) lv = std::max(lv, level[opb[i]] + 1);
    level[i] = lv;
  }
  // sources (constant / instance / witness) are produced as late as possible:
  // one level before their first reader, so they do not occupy a slot early.
  for (size_t i = 0; i < n; ++i) {
    if (absorbed[i]) continue;  // an elided copy reads nothing
    const int ni = n_inputs(tape.kind[i]);
    if (ni >= 1) first_use[opa[i]] = std::min(first_use[opa[i]], level[i]);
    if (ni == 2) first_use[opb[i]] = std::min(first_use[opb[i]], level[i]);
  }
  for (size_t i = 0; i < n; ++i)
    if (n_inputs(tape.kind[i]) == 0 && tape.kind[i] != TK_NOP)
      level[i] = first_use[i] == kInf ? 0 : first_use[i] - 1;
  n_levels = 0;
  for (size_t i = 0; i < n; ++i) {
    if (absorbed[i]) continue;
    const int ni = n_inputs(tape.kind[i]);
    if (ni >= 1) { last_use[opa[i]] = std::max(last_use[opa[i]], level[i]); used[opa[i]] = 1; }
    if (ni == 2) { last_use[opb[i]] = std::max(last_use[opb[i]], level[i]); used[opb[i]] = 1; }
    n_levels = std::max(n_levels, level[i] + 1);
  }
  for (size_t i = 0; i < n; ++i)
    if (!used[i]) last_use[i] = level[i];
  for (uint32_t h : opt.pinned)
    if (h < n) last_use[h] = kInf;
  s.n_levels = n_levels;
}

void ScheduleBuilder::fuse_and_pair() {
  // ---- gate fusion ---------------------------------------------------------
  // An Add/Mul whose value has exactly one reader, itself an Add/Mul, is evaluated inside that reader
  // (depth 1: an op that absorbs cannot be absorbed, an absorbed op has absorbed nothing).  The value is
  // then never materialised, so this is only done when nobody can ask for it afterwards.
  std::vector<uint32_t>& level = s.level_of;
  const bool fuse = opt.fuse && !opt.retain_all && !s.boolean_path;
  pair_second.clear();
  if (fuse) {
    std::vector<uint32_t> reads(n, 0), reader(n, 0), reader0(n, 0);
    std::vector<uint8_t> has_absorbed(n, 0);
    auto note_read = [&](uint32_t p, size_t i) {
      if (reads[p]++ == 0) reader0[p] = (uint32_t)i;
      reader[p] = (uint32_t)i;
    };
    for (size_t i = 0; i < n; ++i) {
      if (absorbed[i]) continue;
      const int ni = n_inputs(tape.kind[i]);
      if (ni >= 1) note_read(opa[i], i);
      if (ni == 2) note_read(opb[i], i);
    }
    auto arith = [&](size_t i) { return tape.kind[i] == TK_ADD || tape.kind[i] == TK_MUL; };
    for (size_t i = 0; i < n; ++i) {
      if (absorbed[i] || !arith(i) || reads[i] != 1 || last_use[i] == kInf || has_absorbed[i]) continue;
      const uint32_t c = reader[i];
      if (!arith(c) || absorbed[c]) continue;
      absorbed[i] = 1;
      has_absorbed[c] = 1;
      ++s.n_absorbed;
      // the producer's operands are now read at the consumer's level
      last_use[opa[i]] = std::max(last_use[opa[i]] == kInf ? kInf : last_use[opa[i]], level[c]);
      last_use[opb[i]] = std::max(last_use[opb[i]] == kInf ? kInf : last_use[opb[i]], level[c]);
    }
    // Shared producers: an Add/Mul read by exactly two Add/Mul gates of one level is evaluated once inside a
    // *pair entry* that produces both readers' values (X = producer; r1 = X o Y -> dst, r2 = X o Z -> dst2).
    // X is never materialised: one store and two loads less.  Y may itself be a fused producer, Z is a plain wire.
    if (opt.pair) {
      pair_second.assign(n, kInf);
      std::vector<uint8_t> in_pair(n, 0);
      for (size_t i = 0; i < n; ++i) {
        if (absorbed[i] || !arith(i) || reads[i] != 2 || last_use[i] == kInf || has_absorbed[i] || in_pair[i]) continue;
        uint32_t c1 = reader0[i], c2 = reader[i];
        if (c1 == c2 || !arith(c1) || !arith(c2) || absorbed[c1] || absorbed[c2] || in_pair[c1] || in_pair[c2] ||
            level[c1] != level[c2])
          continue;
        if (has_absorbed[c2]) std::swap(c1, c2);  // the second gate's other operand must be a plain wire
        if (has_absorbed[c2]) continue;
        absorbed[i] = 3;
        absorbed[c2] = 4;
        pair_second[c1] = c2;
        has_absorbed[c1] = 1;
        in_pair[c1] = in_pair[c2] = 1;
        ++s.n_absorbed;
        ++s.n_paired;
        last_use[opa[i]] = std::max(last_use[opa[i]] == kInf ? kInf : last_use[opa[i]], level[c1]);
        last_use[opb[i]] = std::max(last_use[opb[i]] == kInf ? kInf : last_use[opb[i]], level[c1]);
      }
    }
    s.fused = s.n_absorbed != 0;
  }
}

void ScheduleBuilder::order_by_level() {
  // ---- order ops by (level, kind): counting sort ------------------------
  constexpr uint32_t kKinds = TK_NZ + 1;
  std::vector<uint64_t> bucket((size_t)n_levels * kKinds + 1, 0);
  const std::vector<uint32_t>& level = s.level_of;
  n_live = 0;
  for (size_t i = 0; i < n; ++i)
    if (!absorbed[i]) { ++bucket[(size_t)level[i] * kKinds + tape.kind[i] + 1]; ++n_live; }
  for (size_t k = 1; k < bucket.size(); ++k) bucket[k] += bucket[k - 1];
  order.assign(n_live, 0);
  {
    std::vector<uint64_t> cursor(bucket.begin(), bucket.end() - 1);
    for (size_t i = 0; i < n; ++i)
      if (!absorbed[i]) order[cursor[(size_t)level[i] * kKinds + tape.kind[i]]++] = (uint32_t)i;
  }
  level_start.assign(n_levels + 1, 0);
  for (uint32_t l = 0; l <= n_levels; ++l) level_start[l] = bucket[(size_t)l * kKinds];
}

// wires an op gathers from the table (resolved through elided copies and fused producers)
int ScheduleBuilder::gathered(uint32_t i, uint32_t out[4]) const {
    int k = 0;
    const int ni = n_inputs(tape.kind[i]);
    auto push = [&](uint32_t p) {
      if (absorbed[p] == 1 || absorbed[p] == 3) {
        if (k + 2 <= 4) {
          out[k++] = opa[p];
          out[k++] = opb[p];
        }
      } else if (k < 4) {
        out[k++] = p;
      }
    };
    if (ni >= 1) push(opa[i]);
    if (ni == 2) push(opb[i]);
    return k;  // (a pair entry's fifth wire, the second gate's own operand, is left out of the locality graph)
}

void ScheduleBuilder::assign_slots() {
  // ---- slots: liveness-based reuse, level by level ----------------------
  std::vector<uint32_t> free_slots;
  std::vector<uint32_t> expire_head(n_levels + 1, kInf), expire_next(n, kInf);  // intrusive lists per last_use level
  uint32_t n_slots = 0;
  for (uint32_t l = 0; l < n_levels; ++l) {
    if (!opt.retain_all && l > 0) {
      for (uint32_t h = expire_head[l - 1]; h != kInf; h = expire_next[h]) free_slots.push_back(s.slot_of[h]);
    }
    if (opt.sort_by_operand) {
      // inside a (level, kind) run, order the ops by the slot of their first operand: gates that read
      // the same wire become neighbours (same workgroup), so the repeat read is an L1/L2 hit
      uint64_t k = level_start[l];
      while (k < level_start[l + 1]) {
        uint64_t e = k;
        const uint8_t kind = tape.kind[order[k]];
        while (e < level_start[l + 1] && tape.kind[order[e]] == kind) ++e;
        if (n_inputs(kind) >= 1 && e - k > 1)
          std::stable_sort(order.begin() + k, order.begin() + e, [&](uint32_t x, uint32_t y) {
            auto inner = [&](uint32_t h) { return absorbed[h] == 1 || absorbed[h] == 3; };
            const uint32_t ax = inner(opa[x]) ? opa[opa[x]] : opa[x];
            const uint32_t ay = inner(opa[y]) ? opa[opa[y]] : opa[y];
            return s.slot_of[ax] < s.slot_of[ay];
          });
        k = e;
      }
    }
    if (opt.sort_by_operand >= 2 && !s.boolean_path) {
      // the Add/Mul entries of a level come first (counting sort by kind) and run in a kernel instantiation of
      // their own (engine.hip launch_one): the shared-operand walk orders the two parts separately
      uint64_t mid = level_start[l];
      while (mid < level_start[l + 1] && (tape.kind[order[mid]] == TK_ADD || tape.kind[order[mid]] == TK_MUL)) ++mid;
      const uint64_t cut[3] = {level_start[l], mid, level_start[l + 1]};
      for (int part = 0; part < 2; ++part)
        if (cut[part + 1] - cut[part] > 8)
          locality_order(order.data() + cut[part], cut[part + 1] - cut[part],
                         [&](uint32_t i, uint32_t* out) { return gathered(i, out); });
    }
    for (uint64_t k = level_start[l]; k < level_start[l + 1]; ++k) {
      const uint32_t i = order[k];
      if (tape.kind[i] == TK_ASSERT || tape.kind[i] == TK_NOP) continue;
      uint32_t slot;
      if (!free_slots.empty()) {
        slot = free_slots.back();
        free_slots.pop_back();
      } else {
        slot = n_slots++;
      }
      s.slot_of[i] = slot;
      if (last_use[i] != kInf) {
        expire_next[i] = expire_head[last_use[i]];
        expire_head[last_use[i]] = i;
      }
      if (!pair_second.empty() && pair_second[i] != kInf) {  // the second value of a pair entry needs a slot too
        const uint32_t j = pair_second[i];
        if (!free_slots.empty()) {
          slot = free_slots.back();
          free_slots.pop_back();
        } else {
          slot = n_slots++;
        }
        s.slot_of[j] = slot;
        if (last_use[j] != kInf) {
          expire_next[j] = expire_head[last_use[j]];
          expire_head[last_use[j]] = j;
        }
      }
    }
  }
  s.n_slots = std::max<uint32_t>(n_slots, 1);
}

void ScheduleBuilder::emit_entries() {
  // ---- device ops ----------------------------------------------------------
  if (s.fused) {
    s.ops2.resize(n_live);
    for (size_t k = 0; k < n_live; ++k) {
      const uint32_t i = order[k];
      const uint8_t kind = tape.kind[i];
      DevOp2 d{0, kind, 0, 0, 0, 0, 0, 0};
      d.dst = s.slot_of[i] == kNoWire ? 0 : s.slot_of[i];
      auto operand = [&](uint32_t h, uint32_t* x0, uint32_t* x1, int shift) {
        if (absorbed[h] == 1 || absorbed[h] == 3) {  // evaluated inside this entry (4 = second value of a pair entry: has a slot)
          *x0 = s.slot_of[opa[h]];
          *x1 = s.slot_of[opb[h]];
          d.kind |= (tape.kind[h] == TK_ADD ? 1u : 2u) << shift;
        } else {
          *x0 = s.slot_of[h];
        }
      };
      switch (kind) {
        case TK_ADD: case TK_MUL:
          if (!pair_second.empty() && pair_second[i] != kInf) {
            // pair entry: the shared producer goes first (Add/Mul commute), then the second gate's result slot
            // and its own operand
            const uint32_t j = pair_second[i];
            const uint32_t x = absorbed[opa[i]] == 3 ? opa[i] : opb[i];
            const uint32_t y = absorbed[opa[i]] == 3 ? opb[i] : opa[i];
            operand(x, &d.a0, &d.a1, 8);
            operand(y, &d.b0, &d.b1, 10);
            d.kind |= (tape.kind[j] == TK_ADD ? 1u : 2u) << 12;
            d.pad0 = s.slot_of[j];
            d.pad1 = s.slot_of[opa[j] == x ? opb[j] : opa[j]];
          } else {
            operand(opa[i], &d.a0, &d.a1, 8);
            operand(opb[i], &d.b0, &d.b1, 10);
          }
          break;
        case TK_AND: case TK_XOR:
          d.a0 = s.slot_of[opa[i]];
          d.b0 = s.slot_of[opb[i]];
          break;
        case TK_ADDC: case TK_MULC:
          d.a0 = s.slot_of[opa[i]];
          d.b0 = opb[i];
          break;
        case TK_COPY: case TK_NOT: case TK_NZ: d.a0 = s.slot_of[opa[i]]; break;
        case TK_CONST: case TK_INSTANCE: case TK_WITNESS: d.a0 = opa[i]; break;
        case TK_ASSERT:
          d.a0 = s.slot_of[opa[i]];
          d.b0 = opb[i];
          break;
        default: break;
      }
      s.ops2[k] = d;
    }
  }
  s.ops.resize(s.fused ? 0 : n_live);
  for (size_t k = 0; k < n_live && !s.fused; ++k) {
    const uint32_t i = order[k];
    DevOp d{0, 0, 0, tape.kind[i]};
    const uint8_t kind = tape.kind[i];
    d.dst = s.slot_of[i] == kNoWire ? 0 : s.slot_of[i];
    switch (kind) {
      case TK_ADD: case TK_MUL: case TK_AND: case TK_XOR:
        d.a = s.slot_of[opa[i]];
        d.b = s.slot_of[opb[i]];
        break;
      case TK_ADDC: case TK_MULC:
        d.a = s.slot_of[opa[i]];
        d.b = opb[i];
        break;
      case TK_COPY: case TK_NOT: case TK_NZ: d.a = s.slot_of[opa[i]]; break;
      case TK_CONST: case TK_INSTANCE: case TK_WITNESS: d.a = opa[i]; break;
      case TK_ASSERT:
        d.a = s.slot_of[opa[i]];
        d.b = opb[i];
        break;
      default: break;
    }
    if (s.boolean_path) {  // arithmetic mod 2 on {0,1}: (a+b)%2 = xor, (a*b)%2 = and
      if (kind == TK_ADD) d.kind = TK_XOR;
      else if (kind == TK_MUL) d.kind = TK_AND;
      else if (kind == TK_ADDC) d.kind = const_odd[opb[i]] ? TK_NOT : TK_COPY;
      else if (kind == TK_MULC) {
        if (const_odd[opb[i]]) d.kind = TK_COPY;
        else { d.kind = TK_CONST; d.a = bool_zero_const; }
      }
    }
    s.ops[k] = d;
  }
}

void ScheduleBuilder::emit_launches() {
  // ---- launches -------------------------------------------------------------
  uint32_t l = 0;
  while (l < n_levels) {
    const uint64_t width = level_start[l + 1] - level_start[l];
    s.max_level_width = std::max<uint32_t>(s.max_level_width, (uint32_t)width);
    Launch L;
    L.first = (uint32_t)level_start[l];
    L.level_begin = l;
    if (width >= opt.narrow_width) {
      L.count = (uint32_t)width;
      L.ops_per_wave = 1;
      L.level_end = l + 1;
      ++l;
    } else {
      uint32_t e = l;
      while (e < n_levels && level_start[e + 1] - level_start[e] < opt.narrow_width) ++e;
      L.count = (uint32_t)(level_start[e] - level_start[l]);
      L.ops_per_wave = std::max<uint32_t>(L.count, 1);
      L.sequential = true;
      L.level_end = e;
      l = e;
    }
    for (uint64_t k = L.first; k < (uint64_t)L.first + L.count; ++k) {
      const uint8_t kind = tape.kind[order[k]];
      if (!L.sequential && k == (uint64_t)L.first + L.hot_count && (kind == TK_ADD || kind == TK_MUL)) ++L.hot_count;
      if (!s.boolean_path && (kind == TK_AND || kind == TK_XOR)) L.has_bitops = true;
    }
    s.has_bitops = s.has_bitops || L.has_bitops;
    if (L.count) s.launches.push_back(L);
  }
}

}  // namespace

Schedule build_schedule(const Tape& recorded, const FieldHost& field, const ScheduleOptions& opt) {
  Tape rewritten;
  uint64_t n_ladders = 0;
  const bool use_rewritten = opt.fermat && !opt.retain_all && !field.is_two && rewrite_ladders(recorded, field, &rewritten, &n_ladders);
  const Tape& tape = use_rewritten ? rewritten : recorded;
  ScheduleBuilder b(tape, field, opt);
  Schedule& s = b.s;
  b.n = tape.size();
  s.retain_all = opt.retain_all;
  s.n_ladders = n_ladders;
  s.boolean_path = field.is_two;
  s.slot_of.assign(b.n, kNoWire);
  s.level_of.assign(b.n, 0);
  if (b.n == 0) return s;
  b.device_constants();
  b.propagate_copies();
  b.levelise();
  b.fuse_and_pair();
  b.order_by_level();
  b.assign_slots();
  b.emit_entries();
  b.emit_launches();
  return std::move(b.s);
}

}  // namespace zki
