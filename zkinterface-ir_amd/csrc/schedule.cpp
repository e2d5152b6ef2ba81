#include "schedule.hpp"

#include <unordered_map>

#include "sieve/bignum.hpp"

#include <string.h>

#include <algorithm>
#include <map>
#include <memory>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace zki {
namespace {

inline int n_inputs(uint8_t k) {
  switch (k) {
    case TK_ADD: case TK_MUL: case TK_AND: case TK_XOR: return 2;
    case TK_ADDC: case TK_MULC: case TK_COPY: case TK_NOT: case TK_ASSERT: case TK_NZ: return 1;
    default: return 0;  // CONST, INSTANCE, WITNESS, CARRY, NOP
  }
}

constexpr uint32_t kInf = 0xFFFFFFFFu;
constexpr uint32_t kOperandIsSource = 0x80000000u;   // device/args.hpp: an and / xor operand that names an input, not a slot
constexpr uint32_t kSyntheticOne = 0xFFFFFFFEu;   // GF(2) CONST entry: the pool's synthetic 1 (resolved in finish())
constexpr uint32_t kSyntheticZero = 0xFFFFFFFDu;  // ... and its synthetic 0 (mul_constant by an even constant)

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

// A few worker threads that stay parked between parallel sections: the scheduler has hundreds of short ones per window
// (the runs of every GF(2) level, the levels of an arithmetic window) and starting threads for each would cost more than
// some of them take.  run(n, f): f(0) .. f(n - 1), claimed in index order by the workers and the calling thread; returns
// when all are done.  A task may wait for a task with a SMALLER index (that one has been claimed before it).
class TaskPool {
 public:
  explicit TaskPool(uint32_t workers) {
    for (uint32_t t = 0; t < workers; ++t) threads_.emplace_back([this] { work(); });
  }
  ~TaskPool() {
    {
      std::lock_guard<std::mutex> g(mu_);
      quit_.store(true);
    }
    cv_.notify_all();
    for (auto& th : threads_) th.join();
  }
  uint32_t workers() const { return (uint32_t)threads_.size(); }
  void run(uint32_t n, const std::function<void(uint32_t)>& f) {
    if (n == 0) return;
    if (n == 1 || threads_.empty()) {
      for (uint32_t i = 0; i < n; ++i) f(i);
      return;
    }
    f_ = &f;
    n_ = n;
    pending_.store(n, std::memory_order_relaxed);
    const uint64_t gen = (ticket_.load(std::memory_order_relaxed) >> 32) + 1;
    ticket_.store(gen << 32, std::memory_order_seq_cst);   // publishes f_, n_, pending_: section `gen`, next task 0 (seq_cst: ordered before the look at sleepers_)
    if (sleepers_.load() != 0) {
      std::lock_guard<std::mutex> g(mu_);
      cv_.notify_all();
    }
    drain(gen);
    // the section ends when every task has FINISHED (not just been claimed); sections are short: spin
    while (pending_.load(std::memory_order_acquire) != 0) std::this_thread::yield();
  }

 private:
  // Claim and run tasks of section `gen`.  {section, next task} are ONE word: a claim (the compare-exchange) succeeds only
  // while the section is still current and has an unclaimed task -- and as long as it has one, run() has not returned, so
  // f_ and n_ read between the load and the exchange are that section's.
  void drain(uint64_t gen) {
    for (;;) {
      uint64_t t = ticket_.load(std::memory_order_acquire);
      if ((t >> 32) != gen) return;
      const std::function<void(uint32_t)>* f = f_;
      const uint32_t n = n_;
      const uint32_t i = (uint32_t)t;
      if (i >= n) return;
      if (!ticket_.compare_exchange_weak(t, t + 1, std::memory_order_acq_rel)) continue;
      (*f)(i);
      pending_.fetch_sub(1, std::memory_order_release);
    }
  }
  void work() {
    uint64_t seen = 0;
    auto current = [&] { return ticket_.load(std::memory_order_acquire) >> 32; };
    for (;;) {
      // Sections follow each other within a fraction of a millisecond while a window is scheduled: poll for a moment
      // before going to sleep (waking a parked thread costs about ten microseconds on metal), but not for long: idle
      // pollers take cores from the thread that records the next window
      const auto t0 = std::chrono::steady_clock::now();
      uint32_t polls = 0;
      while (current() == seen && !quit_.load(std::memory_order_relaxed)) {
        if ((++polls & 63) == 0) {
          if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us_)) break;
          std::this_thread::yield();
        }
      }
      if (current() == seen && !quit_.load()) {
        std::unique_lock<std::mutex> lk(mu_);
        sleepers_.fetch_add(1);
        cv_.wait(lk, [&] { return quit_.load() || current() != seen; });
        sleepers_.fetch_sub(1);
      }
      if (quit_.load()) return;
      seen = current();
      drain(seen);
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  const std::function<void(uint32_t)>* f_ = nullptr;
  uint32_t n_ = 0;
  std::atomic<uint64_t> ticket_{0};   // section << 32 | next unclaimed task
  std::atomic<uint32_t> pending_{0}, sleepers_{0};
  std::atomic<bool> quit_{false};
  const long spin_us_ = getenv("ZKI_POOL_SPIN_US") ? atol(getenv("ZKI_POOL_SPIN_US")) : 100;   // (tuning experiments)
};

enum : uint8_t {  // per-op state inside a window
  ST_ENTRY = 0,        // own program entry
  ST_FUSED = 1,        // evaluated inside its only reader
  ST_ELIDED = 2,       // no entry, no slot: propagated copy, dropped ladder op
  ST_PAIR_SHARED = 3,  // shared producer of a pair entry
  ST_PAIR_SECOND = 4,  // second value of a pair entry (has a slot, no entry of its own)
};
enum : uint8_t { FL_DROPPED = 1, FL_PINNED = 2 };

}  // namespace

struct StreamScheduler::Impl {
  FieldHost field;
  ScheduleOptions opt;
  Schedule s;
  uint32_t threads = 1;
  std::unique_ptr<TaskPool> pool_;   // threads - 1 workers, started by the first parallel section
  TaskPool& pool() {
    if (!pool_) pool_.reset(new TaskPool(threads > 1 ? threads - 1 : 0));
    return *pool_;
  }
  // f(i) for i in [0, n) on the pool; f(lo, hi) over [0, n) cut into contiguous slices
  template <class F>
  void parallel_levels(uint32_t n, F&& f) {
    const std::function<void(uint32_t)> fn = f;
    pool().run(n, fn);
  }
  template <class F>
  void parallel_slices(size_t n, size_t min_slice, F&& f) {
    const uint32_t parts = (uint32_t)std::max<size_t>(1, std::min<size_t>(threads, n / std::max<size_t>(min_slice, 1)));
    const std::function<void(uint32_t)> fn = [&](uint32_t p) { f(n * p / parts, n * (p + 1) / parts); };
    pool().run(parts, fn);
  }
  int prime = -1;  // primality of the characteristic: -1 not tested yet

  // per handle, over the whole tape seen so far
  std::vector<uint32_t> root;      // source of the copy chain a propagated copy belongs to (itself otherwise)
  std::vector<uint32_t> last_use;  // global level of the last reader known so far
  std::vector<uint8_t> flags;
  std::vector<uint32_t> open_list;   // values with a slot that may still get readers
  std::vector<uint32_t> free_slots;
  // GF(2): slots are handed out in aligned PAIRS (2d, 2d + 1) -- the two results of a thread of the LDS-resident kernel
  // are one 8-byte store (device/bool_kernels.hpp) -- from free lists per pair bank (d mod 16: the 16 lanes one
  // ds_write_b64 cycle serves hit 16 different bank pairs) when the schedule is bank-aware
  std::vector<std::vector<uint32_t>> free_pairs;    // free pairs per pair bank (one list without bank-awareness)
  std::vector<uint32_t> next_fresh_pair;            // smallest never-used pair of each pair bank
  std::vector<uint8_t> pair_live;                   // per pair: values living in it (0, 1 or 2)
  std::vector<uint64_t> dead_pairs;                 // scratch bitmap of release_batch (assign_slots)
  // strands (runs of narrow levels walked by one workgroup per lane block): per level of the window, one past the last
  // level of its strand (0: the level has a launch of its own), and at a strand's first level the LDS slots it uses
  std::vector<uint32_t> strand_end, strand_lds_slots;
  std::vector<uint32_t> pending_release;            // slots released at the end of the window before (handed back at the start of the next)
  uint32_t open_single = kInf;                      // a pair one half of which went to an op outside the rows
  uint32_t row_pair = kInf;                         // the pair of the even op just placed (its odd neighbour takes the other half)
  uint32_t n_slots = 0;
  uint32_t n_windows = 0;
  std::atomic<uint64_t> dbg_run_ns[4] = {};
  double t_bank_order = 0;   // (ZKI_SCHED_PROFILE) seconds spent ordering GF(2) runs for the LDS banks

  // the window being scheduled (arrays indexed by handle - lo)
  uint32_t lo = 0, hi = 0, base = 0;
  bool final = false;
  std::vector<uint8_t> kind, state;
  std::vector<uint32_t> ra, rb;          // operands resolved through copy chains (handles)
  std::vector<uint32_t> pair_second;     // first gate of a pair entry -> second gate
  std::vector<uint32_t> order;           // live entries in program order
  std::vector<uint64_t> level_start;
  uint32_t n_wlevels = 0;

  bool closed(uint32_t h) const { return (flags[h] & FL_DROPPED) || (final && !(flags[h] & FL_PINNED)); }
  uint8_t& st(uint32_t h) { return state[h - lo]; }
  uint8_t st(uint32_t h) const { return state[h - lo]; }
  bool inner(uint32_t h) const { return h >= lo && (state[h - lo] == ST_FUSED || state[h - lo] == ST_PAIR_SHARED); }

  // Unreduced inputs (see track_unreduced_values): per handle over the whole tape, the source (constant / instance /
  // witness op) whose integer value the handle carries unchanged -- itself for a source, the source of its operand for a
  // copy, kInf for everything an arithmetic gate has produced -- and what the source is
  std::vector<uint32_t> src_root;
  std::vector<uint8_t> src_kind;         // per handle: TK_CONST / TK_INSTANCE / TK_WITNESS for sources, 0 otherwise
  std::vector<uint32_t> src_pos;         // per source handle: input position; a constant: 0 canonical, 1 + its pool index if >= p
  std::map<uint32_t, uint32_t> raw_const_index;   // constant pool index -> position in Schedule::raw_const_of
  std::vector<uint8_t> src_zero_test, src_other;   // per source handle: read by assert_zero / not through copies; by anything else
  std::vector<uint8_t> src_bits;         // per source handle: its raw bits are an operand of and / xor (through copies)
  std::vector<uint32_t> opnd_code[2];    // window: per and / xor op and operand, the input it is a copy of (sink_code's codes), or 0
  std::vector<uint32_t> sink_code;       // window: per op, for assert_zero / not reached from a source through copies alone:
                                         // 1 = the source is a constant >= p, 2 + 2 * position + is_witness = an input
  void grow(uint32_t n);
  void track_unreduced_values(const TapeWindow& w);
  void refuse_unreduced(uint32_t source, const char* consumer);
  void finish_input_modes();
  void rewrite_ladders(const TapeWindow& w);
  void propagate_copies();
  void levelise();
  void fuse_and_pair();
  void place_strand_sources();
  void prefetch_strand_inputs(size_t first_launch);
  std::vector<uint32_t> window_sources;   // constant / instance / witness / carry ops of the window (levelise)
  void order_by_level();
  void assign_slots();
  void order_levels();
  int gathered(uint32_t i, uint32_t out[4]) const;
  void emit_entries();
  void emit_launches();
};

void StreamScheduler::Impl::grow(uint32_t n) {
  const uint32_t old = (uint32_t)root.size();
  if (n <= old) return;
  root.resize(n);
  for (uint32_t i = old; i < n; ++i) root[i] = i;
  last_use.resize(n, 0);
  flags.resize(n, 0);
  s.slot_of.resize(n, kNoWire);
  s.level_of.resize(n, 0);
  src_root.resize(n, kInf);
  src_kind.resize(n, 0);
  src_pos.resize(n, 0);
  src_zero_test.resize(n, 0);
  src_other.resize(n, 0);
  src_bits.resize(n, 0);
}

// PlaintextBackend keeps constants, instance and witness values UNREDUCED (evaluator.rs:862-864,896-898,940-946):
// `copy` clones them as they are, `assert_zero` / `not` test the unreduced integer for zero, `and` / `xor` work on its
// bits, `Evaluator::get` returns it -- while add / mul / add_constant / mul_constant reduce their result (`% m`), and
// over GF(2) the low bit of `a & b` and `a ^ b` only depends on the low bits of a and b.  So a value >= p behaves like
// its residue unless it reaches one of the former through copies alone, and this pass follows every source through its
// copies (forward, over the whole tape: `src_root` outlives the window):
//   * assert_zero / not: a value >= p is NOT zero as an integer, whatever its residue -- decidable, so the sink gets the
//     reference's answer.  GF(p): the sink's entry names the source (`sink_code`) and the kernel tests the raw input
//     beside the wire (fused_entry / replay_kernel).  GF(2): a position read by such sinks alone is packed as `v != 0`
//     instead of `v & 1` (pack_inputs_kernel, mode 0x01); a constant >= p read by such sinks alone becomes the constant 1.
//   * and / xor over an odd field, `Evaluator::get` (a wire alive at the end): the bits of the unreduced integer matter.
//     Refused: per lane for an input >= p at such a position (mode 0xFF, the lane is flagged), at finalize for a
//     constant >= p.  Also refused: the GF(2) position (or constant) that feeds BOTH a zero test and a gate -- one bit
//     cannot be `v & 1` and `v != 0` at once.
// Everything else is reduced on load (to_mont of any value below R is the Montgomery form of its residue).
// The modes are final when the tape has ended (finish_input_modes) and travel in Schedule::strict_instance /
// strict_witness, not in the program entries: a streamed window has been uploaded long before its sources' last readers
// are known, and the verdict must not depend on how the tape was cut.
void StreamScheduler::Impl::refuse_unreduced(uint32_t r, const char* consumer) {
  if (src_kind[r] == TK_CONST) {
    if (src_pos[r])
      throw Error(std::string("GPU backend: a constant >= the field characteristic reaches ") + consumer +
                  " without passing through an arithmetic gate; the reference evaluates that on the unreduced integer "
                  "(evaluator.rs:896-938) and this path does not");
    return;
  }
  std::vector<uint8_t>& pos = src_kind[r] == TK_INSTANCE ? s.strict_instance : src_kind[r] == TK_WITNESS ? s.strict_witness : s.strict_carry;
  if (pos.size() <= src_pos[r]) pos.resize((size_t)src_pos[r] + 1, 0);
  pos[src_pos[r]] = 0xFF;
}

void StreamScheduler::Impl::track_unreduced_values(const TapeWindow& w) {
  const uint32_t n = hi - lo;
  sink_code.assign(n, 0);
  opnd_code[0].assign(n, 0);
  opnd_code[1].assign(n, 0);
  auto code_of = [&](uint32_t r) {
    if (src_kind[r] == TK_CONST) {   // a constant >= p whose bits are read: stream 3, its raw entry in the pool
      const uint32_t c = src_pos[r] - 1;
      auto it = raw_const_index.find(c);
      if (it == raw_const_index.end()) {
        it = raw_const_index.emplace(c, (uint32_t)s.raw_const_of.size()).first;
        s.raw_const_of.push_back(c);
      }
      return 2u + 4u * it->second + 3u;
    }
    return 2u + 4u * src_pos[r] + (src_kind[r] == TK_INSTANCE ? 0u : src_kind[r] == TK_WITNESS ? 1u : 2u);
  };
  for (uint32_t i = lo; i < hi; ++i) {
    const uint8_t k = w.kind[i - lo];
    const uint32_t a = w.a[i - lo], b = w.b[i - lo];
    if (k == TK_CONST || k == TK_INSTANCE || k == TK_WITNESS || k == TK_CARRY) {
      src_root[i] = i;
      src_kind[i] = k;
      src_pos[i] = k == TK_CONST ? (b != 0 ? 1u + a : 0u) : a;   // tape: b != 0 marks a constant that is not canonical
      continue;
    }
    if (k == TK_COPY) {
      src_root[i] = src_root[a];
      continue;
    }
    const int ni = n_inputs(k);
    const bool bit_op = (k == TK_AND || k == TK_XOR) && !field.p_is_two();   // (the low bit of a & b, a ^ b only depends on the low bits)
    for (int q = 0; q < ni; ++q) {
      const uint32_t r = src_root[q == 0 ? a : b];
      if (r == kInf) continue;
      if (k == TK_ASSERT || k == TK_NOT) {
        src_zero_test[r] = 1;
        sink_code[i - lo] = src_kind[r] == TK_CONST ? (src_pos[r] ? 1u : 0u) : code_of(r);
      } else if (bit_op && (src_kind[r] != TK_CONST || src_pos[r])) {
        // the entry reads the raw input (or the raw constant) instead of the wire (device/replay_kernels.hpp bit_operand)
        src_bits[r] = 1;
        opnd_code[q][i - lo] = code_of(r);
      } else {
        src_other[r] = 1;
      }
    }
  }
  if (final && w.pinned)
    for (uint32_t h : *w.pinned) {
      if (h >= hi || src_root[h] == kInf) continue;
      const uint32_t r = src_root[h];
      if (src_kind[r] == TK_CONST && !src_pos[r]) continue;   // a canonical constant: the wire table holds it as it is
      if (opt.pinned_are_carried) {
        // The next field segment takes the value from the wire table, which holds the residue (capi.cpp switch_field
        // re-reads inputs and constants under the new field itself: what is left here is a value carried in and only
        // copied).
        src_other[r] = 1;
        refuse_unreduced(r, "the next field segment (carried over twice without passing through a gate)");
      } else {
        // Evaluator::get returns the integer as it is (evaluator.rs:750-752): zkgpu_get_wire reads the input itself
        src_bits[r] = 1;
        s.raw_source.emplace_back(h, code_of(r));
      }
    }
  if (final) std::sort(s.raw_source.begin(), s.raw_source.end());
}

// after the last window: the input modes of the positions no refusal has claimed, and -- GF(2) -- the sources one bit
// cannot serve
void StreamScheduler::Impl::finish_input_modes() {
  for (uint32_t h = 0; h < (uint32_t)src_kind.size(); ++h) {
    if (!src_kind[h] || !(src_zero_test[h] || src_bits[h])) continue;
    if (src_kind[h] == TK_CONST) {
      if (src_pos[h] && field.is_two && src_other[h])
        refuse_unreduced(h, "assert_zero / not and, as its low bit, a gate");
      continue;
    }
    std::vector<uint8_t>& pos = src_kind[h] == TK_INSTANCE ? s.strict_instance : src_kind[h] == TK_WITNESS ? s.strict_witness : s.strict_carry;
    if (pos.size() <= src_pos[h]) pos.resize((size_t)src_pos[h] + 1, 0);
    uint8_t& mode = pos[src_pos[h]];
    if (mode == 0xFF) continue;
    // 0x01: zero tests only.  0x02 (GF(p)): zero tests and arithmetic -- the kernels treat it like 0 (the sinks test the
    // raw input themselves); the one caller that cares is zkgpu_set_inputs_from_messages with a value wider than the limbs.
    // 0x03: and / xor read its bits (the kernels again read the raw input; a value wider than the limbs flags the lane).
    mode = src_bits[h] ? 0x03 : !src_other[h] ? 0x01 : (field.is_two ? 0xFF : 0x02);
  }
}

// Switch weights are 1 - (case - cond)^(p-1) (evaluator.rs:823-839); the reference computes the power with a
// square-and-multiply ladder (bits(p) squarings + popcount(p-1) multiplies: 352 dependent products at BN254, each of
// them a level of its own).  For a prime p Fermat gives x^(p-1) = 1 for x != 0 and 0 for x = 0, so in the production
// schedule the ladder's result becomes one entry `x != 0` and the ladder's own ops are dropped.  Only done when the
// characteristic passes the primality test (sieve/bignum.cpp), the range IS the reference's recursion, and nothing
// outside the ladder reads or can still read its intermediates.
void StreamScheduler::Impl::rewrite_ladders(const TapeWindow& w) {
  if (!w.n_ladders || !opt.fermat || opt.retain_all || field.p_is_two()) return;
  if (prime < 0) {
    Value p_le;
    for (uint32_t i = 0; i < field.nwords; ++i)
      for (int b = 0; b < 4; ++b) p_le.push_back((uint8_t)(field.p[i] >> (8 * b)));
    prime = is_probably_prime(p_le) ? 1 : 0;
  }
  if (!prime) return;
  // bits of the exponent p - 1 (p is odd: clear bit 0)
  uint32_t e[kFieldWords];
  for (int i = 0; i < kFieldWords; ++i) e[i] = i < (int)field.nwords ? field.p[i] : 0;
  e[0] &= ~1u;
  int top_bit = -1;
  for (int i = 32 * kFieldWords - 1; i >= 0 && top_bit < 0; --i)
    if ((e[i / 32] >> (i % 32)) & 1) top_bit = i;
  if (top_bit < 0) return;
  auto K = [&](uint32_t h) { return w.kind[h - lo]; };
  auto A = [&](uint32_t h) { return w.a[h - lo]; };
  auto B = [&](uint32_t h) { return w.b[h - lo]; };
  // A hint is only a hint (zkgpu_backend_ladder is a public entry): the range must BE the reference's
  // square-and-multiply recursion over p - 1 (evaluator.rs:801-820) -- copy(base), then from the second highest bit
  // down a squaring of the running value and, where the bit is set, a multiply by the base -- or it stays as recorded.
  auto is_ladder = [&](const Tape::Ladder& L) {
    if (L.first < lo || L.result >= hi || L.first > L.result || L.base >= L.first) return false;
    uint32_t cur = L.first;
    if (K(cur) != TK_COPY || A(cur) != L.base) return false;
    for (int shift = top_bit - 1; shift >= 0; --shift) {
      if (++cur > L.result || K(cur) != TK_MUL || A(cur) != cur - 1 || B(cur) != cur - 1) return false;
      if ((e[shift / 32] >> (shift % 32)) & 1) {
        if (++cur > L.result || K(cur) != TK_MUL) return false;
        const bool ab = A(cur) == cur - 1 && B(cur) == L.base, ba = A(cur) == L.base && B(cur) == cur - 1;
        if (!ab && !ba) return false;
      }
    }
    return cur == L.result;
  };
  const uint32_t n = hi - lo;
  std::vector<uint32_t> owner(n, kInf);  // ladder whose intermediate this op is
  std::vector<uint8_t> ok(w.n_ladders, 1);
  for (size_t l = 0; l < w.n_ladders; ++l) {
    const Tape::Ladder& L = w.ladders[l];
    if (!is_ladder(L)) { ok[l] = 0; continue; }
    for (uint32_t i = L.first; i <= L.result; ++i) {
      if (owner[i - lo] != kInf) {  // two hints over one op: neither is trusted, the first keeps the op
        ok[l] = 0;
        ok[owner[i - lo]] = 0;
      } else {
        owner[i - lo] = (uint32_t)l;
      }
      if (i < L.result && !closed(i)) ok[l] = 0;  // a later window could still read the intermediate
    }
  }
  // the result is the one value others may read
  for (size_t l = 0; l < w.n_ladders; ++l) {
    const uint32_t r = w.ladders[l].result;
    if (r >= lo && r < hi && owner[r - lo] == l) owner[r - lo] = kInf;
  }
  for (uint32_t j = lo; j < hi; ++j) {
    const int ni = n_inputs(K(j));
    for (int k = 0; k < ni; ++k) {
      const uint32_t src = k == 0 ? A(j) : B(j);
      if (src < lo || src >= hi || owner[src - lo] == kInf) continue;
      const Tape::Ladder& L = w.ladders[owner[src - lo]];
      if (j < L.first || j > L.result) ok[owner[src - lo]] = 0;  // read from outside the ladder
    }
  }
  for (size_t l = 0; l < w.n_ladders; ++l) {
    if (!ok[l]) continue;
    const Tape::Ladder& L = w.ladders[l];
    for (uint32_t i = L.first; i < L.result; ++i) kind[i - lo] = TK_NOP;
    kind[L.result - lo] = TK_NZ;
    ra[L.result - lo] = L.base;  // resolved through copy chains below, like any operand
    rb[L.result - lo] = 0;
    ++s.n_ladders;
  }
}

void StreamScheduler::Impl::propagate_copies() {
  // The reference's scoping (ingest_subcircuit, evaluator.rs:698-746) copies every input into and every
  // output out of a call / loop body / switch branch: about half of the backend calls of a structured
  // relation are copies.  A copy has the value of its source, so readers are pointed at the source and a
  // copy nobody can observe any more is not materialised.  Copies that must stay
  // readable (retain_all dumps, wires alive at the end, wires that later windows may still read) are kept.
  const bool propagate = opt.propagate_copies && !opt.retain_all;
  for (uint32_t i = lo; i < hi; ++i) {
    const uint8_t k = kind[i - lo];
    if (k == TK_NOP) { st(i) = ST_ELIDED; continue; }  // dropped ladder ops: no entry, no slot, read nothing
    const int ni = n_inputs(k);
    if (ni >= 1) ra[i - lo] = root[ra[i - lo]];
    if (ni == 2) rb[i - lo] = root[rb[i - lo]];
    if (k == TK_COPY && propagate && closed(i)) {
      st(i) = ST_ELIDED;
      root[i] = ra[i - lo];  // source of the whole copy chain
      ++s.n_copies_elided;
    }
  }
}

void StreamScheduler::Impl::levelise() {
  // ---- dependency levels (ASAP), numbered on from the previous window's last level -----------------
  // values of earlier windows are all available before this window's first level
  std::vector<uint32_t>& level = s.level_of;
  auto lvl = [&](uint32_t h) -> int64_t { return h < lo ? (int64_t)base - 1 : (int64_t)level[h]; };
  const uint32_t n = hi - lo;
  // One pass in tape order (operands come before their readers): the level of every op that has operands, and for its
  // operands the last reader known so far and -- sources only -- the first one.  Then the sources: constant / instance /
  // witness values are produced as late as possible, one level before their first reader in the window, so that they do
  // not occupy a slot early.  A value nobody reads lives for its own level.
  std::vector<uint32_t> first_use;   // per source of the window: level of its first reader (allocated when a source shows up)
  std::vector<uint32_t> sources;
  uint32_t top = base;
  auto note_reader = [&](uint32_t x, uint32_t lv) {
    if (last_use[x] < lv) last_use[x] = lv;
    if (x >= lo && n_inputs(kind[x - lo]) == 0 && first_use[x - lo] > lv) first_use[x - lo] = lv;
  };
  for (uint32_t i = lo; i < hi; ++i) {
    if (st(i) == ST_ELIDED) continue;
    const int ni = n_inputs(kind[i - lo]);
    if (ni == 0) {
      if (first_use.empty()) first_use.assign(n, kInf);
      sources.push_back(i);
      level[i] = base;   // provisional (what its readers are levelled against): moved below
      continue;
    }
    int64_t lv = lvl(ra[i - lo]) + 1;
    if (ni == 2) lv = std::max(lv, lvl(rb[i - lo]) + 1);
    level[i] = (uint32_t)lv;
    if (last_use[i] < (uint32_t)lv) last_use[i] = (uint32_t)lv;
    note_reader(ra[i - lo], (uint32_t)lv);
    if (ni == 2) note_reader(rb[i - lo], (uint32_t)lv);
    top = std::max(top, (uint32_t)lv + 1);
  }
  for (uint32_t i : sources) {
    level[i] = first_use[i - lo] == kInf ? base : first_use[i - lo] - 1;
    if (last_use[i] < level[i]) last_use[i] = level[i];
    top = std::max(top, level[i] + 1);
  }
  window_sources.swap(sources);
  n_wlevels = top - base;
}

void StreamScheduler::Impl::fuse_and_pair() {
  // ---- gate fusion ---------------------------------------------------------
  // An Add/Mul whose value has exactly one reader, itself an Add/Mul, is evaluated inside that reader
  // (depth 1: an op that absorbs cannot be absorbed, an absorbed op has absorbed nothing).  The value is
  // then never materialised, so this is only done when nobody can ask for it afterwards: the value is closed
  // (dropped by its owner, or the tape has ended and it is not pinned) -- its readers are then all in this window.
  const std::vector<uint32_t>& level = s.level_of;
  const uint32_t n = hi - lo;
  pair_second.clear();
  if (!s.fused) return;   // (opt.fuse, no retain_all, a Montgomery field)
  std::vector<uint32_t> reads(n, 0), reader(n, 0), reader0(n, 0);
  std::vector<uint8_t> has_absorbed(n, 0);
  auto note_read = [&](uint32_t p, uint32_t i) {
    if (p < lo) return;
    if (reads[p - lo]++ == 0) reader0[p - lo] = i;
    reader[p - lo] = i;
  };
  for (uint32_t i = lo; i < hi; ++i) {
    if (st(i) != ST_ENTRY) continue;
    const int ni = n_inputs(kind[i - lo]);
    if (ni >= 1) note_read(ra[i - lo], i);
    if (ni == 2) note_read(rb[i - lo], i);
  }
  auto arith = [&](uint32_t i) { return kind[i - lo] == TK_ADD || kind[i - lo] == TK_MUL; };
  auto extend = [&](uint32_t h, uint32_t lv) { last_use[h] = std::max(last_use[h], lv); };
  for (uint32_t i = lo; i < hi; ++i) {
    if (st(i) != ST_ENTRY || !arith(i) || reads[i - lo] != 1 || !closed(i) || has_absorbed[i - lo]) continue;
    const uint32_t c = reader[i - lo];
    if (!arith(c) || st(c) != ST_ENTRY) continue;
    st(i) = ST_FUSED;
    has_absorbed[c - lo] = 1;
    ++s.n_absorbed;
    // the producer's operands are now read at the consumer's level
    extend(ra[i - lo], level[c]);
    extend(rb[i - lo], level[c]);
  }
  // Shared producers: an Add/Mul read by exactly two Add/Mul gates of one level is evaluated once inside a
  // *pair entry* that produces both readers' values (X = producer; r1 = X o Y -> dst, r2 = X o Z -> dst2).
  // X is never materialised: one store and two loads less.  Y may itself be a fused producer, Z is a plain wire.
  if (opt.pair) {
    pair_second.assign(n, kInf);
    std::vector<uint8_t> in_pair(n, 0);
    for (uint32_t i = lo; i < hi; ++i) {
      if (st(i) != ST_ENTRY || !arith(i) || reads[i - lo] != 2 || !closed(i) || has_absorbed[i - lo] || in_pair[i - lo]) continue;
      uint32_t c1 = reader0[i - lo], c2 = reader[i - lo];
      if (c1 == c2 || !arith(c1) || !arith(c2) || st(c1) != ST_ENTRY || st(c2) != ST_ENTRY || in_pair[c1 - lo] ||
          in_pair[c2 - lo] || level[c1] != level[c2])
        continue;
      if (has_absorbed[c2 - lo]) std::swap(c1, c2);  // the second gate's other operand must be a plain wire
      if (has_absorbed[c2 - lo]) continue;
      st(i) = ST_PAIR_SHARED;
      st(c2) = ST_PAIR_SECOND;
      pair_second[c1 - lo] = c2;
      has_absorbed[c1 - lo] = 1;
      in_pair[c1 - lo] = in_pair[c2 - lo] = 1;
      ++s.n_absorbed;
      ++s.n_paired;
      extend(ra[i - lo], level[c1]);
      extend(rb[i - lo], level[c1]);
    }
  }
}

// Inside a STRAND (a run of narrow levels walked by one workgroup, a barrier per level) "one level before its first reader"
// can put an input's fetch from HBM and its conversion -- a whole Montgomery product -- into a level of the chain that only
// adds (that level then lasts as long as the input takes: measured on the chained structured relation, 4,100 cycles instead
// of 1,500), or into a level whose gates have all been absorbed by their readers, which would otherwise not exist at all.
// Once fusion has decided what the levels hold, an input of a narrow region moves up to the nearest level in front of its
// reader that has work of its own, products if possible, and a wave to spare.
void StreamScheduler::Impl::place_strand_sources() {
  if (!s.fused || window_sources.empty() || !n_wlevels) return;
  std::vector<uint32_t>& level = s.level_of;
  std::vector<uint16_t> count(n_wlevels, 0), heavy(n_wlevels, 0);   // entries per level (saturating), entries with a product
  auto bump = [](uint16_t& c) { if (c < 0xFFFF) ++c; };
  auto has_product = [&](uint32_t i) {
    const uint8_t k = kind[i - lo];
    if (k == TK_MUL || k == TK_MULC || k == TK_INSTANCE || k == TK_WITNESS || k == TK_CARRY) return true;
    if (k != TK_ADD) return false;
    for (uint32_t x : {ra[i - lo], rb[i - lo]})
      if (inner(x) && kind[x - lo] == TK_MUL) return true;
    return false;
  };
  for (uint32_t i = lo; i < hi; ++i) {
    if (st(i) != ST_ENTRY || n_inputs(kind[i - lo]) == 0) continue;
    bump(count[level[i] - base]);
    if (has_product(i)) bump(heavy[level[i] - base]);
  }
  const uint32_t narrow = std::max(opt.strand_width, opt.narrow_width);
  for (uint32_t i : window_sources) {
    if (st(i) != ST_ENTRY) continue;
    const uint32_t P = level[i] - base;
    uint32_t target = P;
    if (count[P] < narrow) {
      const bool converts = kind[i - lo] != TK_CONST;
      const uint32_t floor_l = P >= 4 ? P - 4 : 0;
      uint32_t nonempty = kInf, good = kInf;
      for (uint32_t L = P + 1; L-- > floor_l;) {
        if (count[L] >= narrow) break;   // (a wide level: not part of this strand)
        if (count[L] == 0) continue;
        if (nonempty == kInf) nonempty = L;
        if (converts && heavy[L] > 0 && count[L] < 4) { good = L; break; }
      }
      target = good != kInf ? good : nonempty != kInf ? nonempty : P;
    }
    level[i] = base + target;
    bump(count[target]);
    if (kind[i - lo] != TK_CONST) bump(heavy[target]);
  }
}

void StreamScheduler::Impl::order_by_level() {
  // ---- order ops by (level, kind): counting sort ------------------------
  // GF(2): the kinds that run as ROWS of the LDS-resident kernel (and, xor, not, copy) come last in a level, and first
  // then xor, not, copy -- for the kernel the last three are all `xor` (with ONES / ZERO as the second operand): the rows
  // of a level are ONE sequence of ops whose first part is `and` and whose rest is `xor`, padded at its end only
  // (lds_program.cpp; positions in that sequence: assign_slots).
  constexpr uint32_t kKinds = TK_CARRY + 1;
  uint32_t rank[kKinds];
  for (uint32_t k = 0; k < kKinds; ++k) rank[k] = k;
  if (s.boolean_path) {
    uint32_t next = 0;
    for (uint32_t k = 0; k < kKinds; ++k)
      if (k != TK_AND && k != TK_XOR && k != TK_NOT && k != TK_COPY) rank[k] = next++;
    for (uint32_t k : {(uint32_t)TK_AND, (uint32_t)TK_XOR, (uint32_t)TK_NOT, (uint32_t)TK_COPY}) rank[k] = next++;
  }
  std::vector<uint64_t> bucket((size_t)n_wlevels * kKinds + 1, 0);
  const std::vector<uint32_t>& level = s.level_of;
  size_t n_live = 0;
  for (uint32_t i = lo; i < hi; ++i)
    if (st(i) == ST_ENTRY) { ++bucket[(size_t)(level[i] - base) * kKinds + rank[kind[i - lo]] + 1]; ++n_live; }
  for (size_t k = 1; k < bucket.size(); ++k) bucket[k] += bucket[k - 1];
  order.assign(n_live, 0);
  {
    std::vector<uint64_t> cursor(bucket.begin(), bucket.end() - 1);
    for (uint32_t i = lo; i < hi; ++i)
      if (st(i) == ST_ENTRY) order[cursor[(size_t)(level[i] - base) * kKinds + rank[kind[i - lo]]]++] = i;
  }
  level_start.assign(n_wlevels + 1, 0);
  for (uint32_t l = 0; l <= n_wlevels; ++l) level_start[l] = bucket[(size_t)l * kKinds];
}

// wires an op gathers from the table (resolved through elided copies and fused producers)
int StreamScheduler::Impl::gathered(uint32_t i, uint32_t out[4]) const {
  int k = 0;
  const int ni = n_inputs(kind[i - lo]);
  auto push = [&](uint32_t p) {
    if (inner(p)) {
      if (k + 2 <= 4) {
        out[k++] = ra[p - lo];
        out[k++] = rb[p - lo];
      }
    } else if (k < 4) {
      out[k++] = p;
    }
  };
  if (ni >= 1) push(ra[i - lo]);
  if (ni == 2) push(rb[i - lo]);
  return k;  // (a pair entry's fifth wire, the second gate's own operand, is left out of the locality graph)
}

void StreamScheduler::Impl::assign_slots() {
  // ---- slots: liveness-based reuse, level by level ----------------------
  // A slot is reused only by a strictly later level than its last reader, so a level never overwrites what it reads.
  // A value keeps its slot while it is open (its owner has not dropped it): later windows may read it.
  std::vector<uint32_t> expire_head(n_wlevels + 1, kInf);
  std::vector<uint32_t> expire_next(hi - lo, kInf);  // intrusive lists per last_use level, this window's values
  struct Ext {
    uint32_t h, next;
  };
  std::vector<Ext> ext;                               // the same for values of earlier windows
  std::vector<uint32_t> ext_head(n_wlevels + 1, kInf);
  std::vector<uint32_t> free_now;
  if (!opt.retain_all) {
    // values of earlier windows that were still open: closed now?
    size_t keep = 0;
    for (uint32_t h : open_list) {
      if (!closed(h)) { open_list[keep++] = h; continue; }
      if (last_use[h] < base || n_wlevels == 0) {
        free_now.push_back(s.slot_of[h]);     // (handed back below, once the allocator of the field is at hand)
      } else {
        ext.push_back({h, ext_head[last_use[h] - base]});
        ext_head[last_use[h] - base] = (uint32_t)ext.size() - 1;
      }
    }
    open_list.resize(keep);
  }
  // GF(2): the LDS-resident kernel keeps a slice's wire table in LDS (device/bool_kernels.hpp), where what costs is
  // bank conflicts -- 32 lanes gathering random slots hit some bank 3 to 4 times.  The scheduler owns both the order
  // of a level's ops and the slot numbering, so it can make the accesses of one LDS instruction hit 32 different
  // banks: `banked` (below) orders the ops of a (level, kind) run so that the 32 ops one `ds_read_b32` group serves
  // read operands from different banks, and gives the op at lane k a result slot in bank k mod 32.
  const bool banked = s.boolean_path && opt.bank_aware && !opt.retain_all;
  constexpr uint32_t kBanks = 32, kPairBanks = 16;
  const uint32_t n_pair_lists = banked ? kPairBanks : 1;
  if (s.boolean_path && free_pairs.empty()) {
    free_pairs.resize(n_pair_lists);
    next_fresh_pair.resize(n_pair_lists);
    for (uint32_t b = 0; b < n_pair_lists; ++b) next_fresh_pair[b] = b;
  }
  // a free pair of the wanted pair bank (kInf: any -- the fullest free list, else the smallest pair never used)
  auto take_pair = [&](uint32_t bank) {
    if (bank == kInf || !banked) {
      uint32_t best = kInf;
      for (uint32_t b = 0; b < n_pair_lists; ++b)
        if (!free_pairs[b].empty() && (best == kInf || free_pairs[b].size() > free_pairs[best].size())) best = b;
      if (best == kInf)
        for (uint32_t b = 0; b < n_pair_lists; ++b)
          if (best == kInf || next_fresh_pair[b] < next_fresh_pair[best]) best = b;
      bank = best;
    }
    uint32_t d;
    if (!free_pairs[bank].empty()) {
      d = free_pairs[bank].back();
      free_pairs[bank].pop_back();
    } else {
      d = next_fresh_pair[bank];
      next_fresh_pair[bank] += n_pair_lists;
    }
    if (pair_live.size() <= d) pair_live.resize((size_t)d + 1, 0);
    n_slots = std::max(n_slots, 2 * d + 2);
    return d;
  };
  // GF(2).  position: kInf = an op outside the rows (inputs, constants, narrow levels: one half of a shared pair);
  // otherwise the op's position in its (level, kind) run -- the even op of a thread opens a pair of the thread's pair
  // bank, its odd neighbour takes the other half.
  bool prefer_bank = false;   // the run being placed is long enough for whole groups of lanes: its threads get their pair banks
  auto take_bool_slot = [&](uint32_t position) {
    uint32_t slot;
    if (position == kInf) {
      if (open_single == kInf) {
        open_single = take_pair(kInf);
        slot = 2 * open_single;
      } else {
        slot = 2 * open_single + 1;
        open_single = kInf;
      }
    } else if ((position & 1) == 0) {
      row_pair = take_pair(banked && prefer_bank ? (position / 2) % kPairBanks : kInf);
      slot = 2 * row_pair;
    } else {
      slot = 2 * row_pair + 1;
    }
    ++pair_live[slot / 2];
    return slot;
  };
  auto take_slot = [&](uint32_t position) {
    if (s.boolean_path) return take_bool_slot(position);
    if (!free_slots.empty()) {
      const uint32_t slot = free_slots.back();
      free_slots.pop_back();
      return slot;
    }
    return n_slots++;
  };
  // Slots that become free together (the values whose last reader sits in one level; at the start of a window also
  // what the window before left) are handed back as ONE batch.  GF(2): the pairs of a batch that are dead in both halves
  // go onto the free lists in ascending order whatever order the values came in -- a window that closes values of the
  // window before it finds them on other lists than the one-window schedule does, and the slot numbers (the bytes of the
  // LDS program) must not depend on that.
  // Strands: what a strand produces and reads for the last time inside itself, and nobody can ask for afterwards, lives in
  // the LDS of the workgroup that walks it (kSlotInLds) -- a strand's temporaries are re-read by the same workgroup one
  // level later, and through the wire table that is a store, its acknowledgement and a load from L2 per level.
  const uint32_t narrow_w = s.fused ? std::max(opt.strand_width, opt.narrow_width) : 0;
  strand_end.assign(n_wlevels, 0);
  strand_lds_slots.assign(n_wlevels, 0);
  if (narrow_w && !opt.retain_all && opt.strand_lds)
    for (uint32_t l = 0; l < n_wlevels;) {
      if (level_start[l + 1] - level_start[l] >= narrow_w) { ++l; continue; }
      uint32_t e = l;
      while (e < n_wlevels && level_start[e + 1] - level_start[e] < narrow_w) ++e;
      for (uint32_t q = l; q < e; ++q) strand_end[q] = e;
      l = e;
    }
  const uint32_t lds_value_bytes = ((field.nwords + 3) / 4) * 64 * 16;
  const uint32_t lds_cap = std::min<uint32_t>(1024, kStrandLdsBytes / std::max<uint32_t>(lds_value_bytes, 1));
  std::vector<uint32_t> lds_free;
  uint32_t lds_next = 0, strand_first = 0;
  std::vector<uint32_t> batch;
  auto release_batch = [&]() {
    if (!s.boolean_path) {
      for (uint32_t slot : batch)
        if (slot & kSlotInLds) lds_free.push_back(slot & ~kSlotInLds);
        else free_slots.push_back(slot);
      batch.clear();
      return;
    }
    uint32_t w_lo = kInf, w_hi = 0;
    for (uint32_t slot : batch) {
      const uint32_t d = slot / 2;
      if (--pair_live[d] != 0 || d == open_single) continue;   // (the other half is alive, or still to be handed out)
      if (dead_pairs.size() <= d / 64) dead_pairs.resize(d / 64 + 1, 0);
      dead_pairs[d / 64] |= 1ull << (d % 64);
      w_lo = std::min(w_lo, d / 64);
      w_hi = std::max(w_hi, d / 64);
    }
    batch.clear();
    for (uint32_t w = w_lo; w != kInf && w <= w_hi; ++w)
      for (uint64_t bits = dead_pairs[w]; bits; bits &= bits - 1) {
        const uint32_t d = w * 64 + (uint32_t)__builtin_ctzll(bits);
        free_pairs[banked ? d % kPairBanks : 0].push_back(d);
      }
    for (uint32_t w = w_lo; w != kInf && w <= w_hi; ++w) dead_pairs[w] = 0;
  };
  // what the window before this one released at its end + its values that have been closed since
  batch.swap(pending_release);
  batch.insert(batch.end(), free_now.begin(), free_now.end());
  release_batch();
  // Order the ops of one (level, kind) run [k0, k1) of `order` for the LDS kernel.  The row ops of a level are ONE
  // sequence (and, then xor, not, copy: order_by_level) and thread t of the workgroup executes ops 2t and 2t + 1 of a
  // 2048-op row of it, so one LDS instruction of a wave serves the even (or the odd) ops of a 128-op block, in two groups
  // of 32 lanes: positions q and q' of the SEQUENCE conflict when q / 64 == q' / 64, q % 2 == q' % 2 and their operands
  // (or their results) share a bank.  `offset` = position of the run's first op in the sequence: the run's first and
  // last 64-position blocks may be shared with its neighbours, whose banks in the shared block arrive in carry_a / carry_b
  // (per parity) and are avoided.  Greedy: fill group after group, taking for each lane an op whose operand banks are still
  // unused in the group; what cannot be placed conflict-free fills the holes.
  // The runs of a level are ordered CONCURRENTLY (they are independent but for the 64-position block two neighbours may
  // share): a run first orders the groups of the blocks it owns alone, then waits for its predecessor's banks in the shared
  // block (`Carry`, per parity) and orders its own part of that block last.
  struct Carry {
    uint32_t a[2] = {0, 0}, b[2] = {0, 0};
  };
  // after a run has its final order: the banks its ops use in the block the next run will share with it (`c` holds the
  // carry the run itself started from)
  auto note_run = [&](uint64_t k0, uint64_t k1, uint64_t offset, Carry& c) {
    const uint64_t end = offset + (k1 - k0);
    if (k1 == k0) return;
    if (end % 64 == 0) {
      c = Carry();
      return;
    }
    const uint64_t last_block = (end - 1) / 64;
    if (offset / 64 != last_block || offset % 64 == 0) c = Carry();   // (else: the run lies inside the shared block)
    for (uint64_t q = std::max(offset, last_block * 64); q < end; ++q) {
      const uint32_t i = order[k0 + (q - offset)];
      const int ni = n_inputs(kind[i - lo]);
      if (ni >= 1) c.a[q % 2] |= 1u << (s.slot_of[ra[i - lo]] % kBanks);
      if (ni == 2) c.b[q % 2] |= 1u << (s.slot_of[rb[i - lo]] % kBanks);
    }
  };
  // group order of a run: the groups of its first block last when that block is shared with the run before it
  auto group_at = [](size_t gi, size_t n_groups, bool shared_first) { return !shared_first ? gi : (gi + 2 < n_groups ? gi + 2 : gi + 2 - n_groups); };
  auto bank_order = [&](uint64_t k0, uint64_t k1, uint64_t offset, const std::function<const Carry&()>& wait_carry) {
    const size_t cnt = k1 - k0;
    if (cnt < 2 * kBanks) return;
    const bool two = n_inputs(kind[order[k0] - lo]) == 2;
    std::vector<uint32_t> run(order.begin() + k0, order.begin() + k1);
    std::vector<std::vector<uint32_t>> by_a(kBanks);
    for (uint32_t i : run) {
      // and / xor commute: take as operand a the one whose bank has fewer ops so far (two choices per op keep the 32
      // buckets within a few ops of each other, so the groups stay conflict-free until the run is almost used up)
      if (two) {
        const uint32_t A = s.slot_of[ra[i - lo]] % kBanks, B = s.slot_of[rb[i - lo]] % kBanks;
        if (by_a[A].size() > by_a[B].size()) std::swap(ra[i - lo], rb[i - lo]);
      }
      by_a[s.slot_of[ra[i - lo]] % kBanks].push_back(i);
    }
    std::vector<uint32_t> out(cnt, kInf);
    const uint64_t block0 = offset / 64;
    const size_t n_groups = (size_t)((offset + cnt - 1) / 64 - block0 + 1) * 2;
    // groups in sequence order: block block0 + g / 2, parity g % 2 -> sequence positions block * 64 + (g % 2) + 2 * lane;
    // the run holds the positions [offset, offset + cnt): index into `out` = position - offset, kInf outside
    auto pos_of = [&](size_t g, uint32_t lane) -> size_t {
      const uint64_t q = (block0 + g / 2) * 64 + (g % 2) + 2 * (uint64_t)lane;
      return (q < offset || q >= offset + cnt) ? (size_t)kInf : (size_t)(q - offset);
    };
    std::vector<uint32_t> used_a(n_groups, 0), used_b(n_groups, 0);
    const bool shared_first = offset % 64 != 0;
    // ops left per operand-b bank: a lane prefers the candidate whose b bank has most ops left, so that the banks
    // are used up evenly and the last groups of the run still find 32 different ones
    uint32_t left_b[kBanks] = {0};
    if (two)
      for (uint32_t i : run) ++left_b[s.slot_of[rb[i - lo]] % kBanks];
    // take from bucket `bank` an op whose operand-b bank is free in the group (looking at the last `look` ops)
    auto take = [&](uint32_t bank, uint32_t ub, size_t look) {
      std::vector<uint32_t>& bucket = by_a[bank];
      look = std::min(look, bucket.size());
      size_t best = (size_t)-1;
      uint32_t best_left = 0;
      for (size_t c = 0; c < look; ++c) {
        const uint32_t i = bucket[bucket.size() - 1 - c];
        if (!two) { best = c; break; }
        const uint32_t bb = s.slot_of[rb[i - lo]] % kBanks;
        if (!((ub >> bb) & 1) && left_b[bb] > best_left) {
          best = c;
          best_left = left_b[bb];
        }
      }
      if (best == (size_t)-1) return kInf;
      const uint32_t i = bucket[bucket.size() - 1 - best];
      std::swap(bucket[bucket.size() - 1 - best], bucket.back());
      bucket.pop_back();
      if (two) --left_b[s.slot_of[rb[i - lo]] % kBanks];
      return i;
    };
    auto put = [&](size_t g, uint32_t lane, uint32_t i) {
      out[pos_of(g, lane)] = i;
      used_a[g] |= 1u << (s.slot_of[ra[i - lo]] % kBanks);
      if (two) used_b[g] |= 1u << (s.slot_of[rb[i - lo]] % kBanks);
    };
    for (size_t gi = 0; gi < n_groups; ++gi) {
      const size_t g = group_at(gi, n_groups, shared_first);
      if (shared_first && g == 0) {   // the banks the run before this one uses in the block the two share
        const Carry& c = wait_carry();
        for (int par = 0; par < 2; ++par) {
          used_a[par] = c.a[par];
          used_b[par] = c.b[par];
        }
      }
      uint32_t holes[kBanks], n_holes = 0;
      for (uint32_t lane = 0; lane < kBanks; ++lane) {
        if (pos_of(g, lane) == (size_t)kInf) continue;
        // the operand-a bank of this lane: rotate with the group so that no bank's bucket is always served last
        const uint32_t bank = (lane + (uint32_t)g) % kBanks;
        const uint32_t i = ((used_a[g] >> bank) & 1) ? kInf : take(bank, used_b[g], 24);
        if (i != kInf) put(g, lane, i);
        else holes[n_holes++] = lane;
      }
      // lanes still empty: any bucket whose bank the group does not read yet, looking deeper for a free b bank
      for (uint32_t h = 0; h < n_holes; ++h) {
        for (uint32_t bank = 0; bank < kBanks; ++bank) {
          if ((used_a[g] >> bank) & 1) continue;
          const uint32_t i = take(bank, used_b[g], 512);
          if (i != kInf) { put(g, holes[h], i); break; }
        }
      }
    }
    // what is left cannot be placed without a conflict: fill the remaining holes, a free operand-a bank first
    std::vector<uint8_t> hits(n_groups * kBanks, 0);   // per group and operand-a bank: ops placed by this pass
    for (size_t g = 0; g < n_groups; ++g)
      for (uint32_t lane = 0; lane < kBanks; ++lane) {
        const size_t q = pos_of(g, lane);
        if (q == (size_t)kInf || out[q] != kInf) continue;
        uint32_t pick_bank = kInf;
        for (uint32_t bank = 0; bank < kBanks && pick_bank == kInf; ++bank)
          if (!by_a[bank].empty() && !((used_a[g] >> bank) & 1)) pick_bank = bank;
        // (no free bank left: the one this group has hit least often so far)
        if (pick_bank == kInf)
          for (uint32_t bank = 0; bank < kBanks; ++bank)
            if (!by_a[bank].empty() && (pick_bank == kInf || hits[g * kBanks + bank] < hits[g * kBanks + pick_bank])) pick_bank = bank;
        const uint32_t i = by_a[pick_bank].back();
        by_a[pick_bank].pop_back();
        ++hits[g * kBanks + pick_bank];
        put(g, lane, i);
      }
    for (size_t q = 0; q < cnt; ++q) order[k0 + q] = out[q];
  };
  // The same for a run of two-operand ops (and / xor), where a group needs 32 ops with 32 different operand-a banks AND
  // 32 different operand-b banks: a perfect matching in the bipartite graph a-bank -- b-bank whose edges are the ops
  // still unplaced (an op with operand banks (x, y) is the edge x--y and, the gates being commutative, also y--x with
  // its operands swapped).  Greedy choices run dry towards the end of a run; augmenting paths (Kuhn) do not, as long
  // as a matching exists.
  // `out` holds positions in `run` and operand swaps are only NOTED (bank_xy swapped, the op listed in `flips`): the runs of
  // a level are ordered concurrently and their ops lie interleaved in ra / rb -- writing those from several threads makes
  // the cache lines bounce (measured: every run took twice as long).  The caller swaps the listed ops' operands afterwards.
  // carry_out: the banks the run uses in its last 64-position block (note_run, from the run's own bank table).
  auto bank_order_two = [&](uint64_t k0, uint64_t k1, uint64_t offset, const std::function<const Carry&()>& wait_carry,
                            std::vector<uint32_t>& flips, Carry& carry_out) {
    const size_t cnt = k1 - k0;
    constexpr uint32_t kFlip = 0x80000000u;
    std::vector<uint32_t> run(order.begin() + k0, order.begin() + k1);
    // (a bank, b bank) -> positions in `run` (| kFlip: operands swapped), as one array cut by cell_end: the entries of cell
    // c are cell_item[cell_begin[c] .. cell_end[c]), consumed from the back (two counting passes instead of a thousand
    // small vectors per run)
    std::vector<uint32_t> cell_begin(kBanks * kBanks + 1, 0), cell_end(kBanks * kBanks, 0), cell_item(2 * cnt);
    std::vector<uint16_t> bank_xy(cnt);
    for (size_t r = 0; r < cnt; ++r) {
      const uint32_t i = run[r], x = s.slot_of[ra[i - lo]] % kBanks, y = s.slot_of[rb[i - lo]] % kBanks;
      bank_xy[r] = (uint16_t)(x * kBanks + y);
      ++cell_begin[x * kBanks + y + 1];
      if (x != y) ++cell_begin[y * kBanks + x + 1];
    }
    for (uint32_t c = 0; c < kBanks * kBanks; ++c) cell_begin[c + 1] += cell_begin[c];
    for (uint32_t c = 0; c < kBanks * kBanks; ++c) cell_end[c] = cell_begin[c];
    for (size_t r = 0; r < cnt; ++r) {
      const uint32_t x = bank_xy[r] / kBanks, y = bank_xy[r] % kBanks;
      cell_item[cell_end[x * kBanks + y]++] = (uint32_t)r;
      if (x != y) cell_item[cell_end[y * kBanks + x]++] = (uint32_t)r | kFlip;
    }
    std::vector<uint8_t> used(cnt, 0);
    // unplaced ops per edge (symmetric: x--y and y--x are the same ops)
    std::vector<uint32_t> left(kBanks * kBanks, 0);
    for (uint32_t c = 0; c < kBanks * kBanks; ++c) left[c] = cell_end[c] - cell_begin[c];
    std::vector<uint32_t> out(cnt, kInf);
    const uint64_t block0 = offset / 64;
    const size_t n_groups = (size_t)((offset + cnt - 1) / 64 - block0 + 1) * 2;
    auto pos_of = [&](size_t g, uint32_t lane) -> size_t {   // (as in bank_order)
      const uint64_t q = (block0 + g / 2) * 64 + (g % 2) + 2 * (uint64_t)lane;
      return (q < offset || q >= offset + cnt) ? (size_t)kInf : (size_t)(q - offset);
    };
    // One group's matching state.  `avail[a]`: the b banks with ops left on edge a--b, as a bit mask -- the scan for a free
    // b bank only visits those (this loop is the scheduler's hot spot: 330,000 groups of 32 lanes for the C4 relation).
    struct Matcher {
      int match_of_b[kBanks], match_of_a[kBanks];
      uint32_t visited = 0, taken_b = 0;
      uint32_t choices = 0, rotation = 0;   // choices: 0 = look at every free b bank
      const uint32_t* left;
      const uint32_t* avail;
      // capacity of edge a--b for one more use in this group: the reverse edge b--a, if matched, draws on the same ops
      inline uint32_t capacity(uint32_t a, uint32_t b) const {
        const uint32_t n = left[a * kBanks + b];
        return (a != b && match_of_a[b] == (int)a) ? (n > 1 ? n - 1 : 0) : n;
      }
      bool augment(uint32_t a) {
        // a free b bank if there is one: the edge with most ops left, so that the edges are used up evenly and the graph
        // stays dense to the end of the run
        uint32_t best = kBanks, best_n = 0;
        const uint32_t free_b = avail[a] & ~taken_b;
        if (choices && free_b) {
          // a few candidates, starting at a bank that rotates with the group and with a: the one with most ops left
          const uint32_t rot = (rotation + 7 * a) & 31;
          uint32_t cand = (free_b >> rot) | (rot ? free_b << (32 - rot) : 0);
          for (uint32_t c = 0; c < choices && cand; ++c, cand &= cand - 1) {
            const uint32_t b = ((uint32_t)__builtin_ctz(cand) + rot) & 31;
            const uint32_t n = capacity(a, b);
            if (n > best_n) { best_n = n; best = b; }
          }
        }
        if (best == kBanks)
          for (uint32_t cand = free_b; cand; cand &= cand - 1) {
            const uint32_t b = (uint32_t)__builtin_ctz(cand);
            const uint32_t n = capacity(a, b);
            if (n > best_n) { best_n = n; best = b; }
          }
        if (best < kBanks) {
          match_of_b[best] = (int)a;
          match_of_a[a] = (int)best;
          taken_b |= 1u << best;
          return true;
        }
        // otherwise push somebody else off a taken one (augmenting path); (taken without a match: by the neighbouring run)
        for (uint32_t cand = avail[a] & taken_b & ~visited; cand; cand &= cand - 1) {
          const uint32_t b = (uint32_t)__builtin_ctz(cand);
          if ((visited >> b) & 1 || match_of_b[b] < 0 || !capacity(a, b)) continue;
          visited |= 1u << b;
          const int other = match_of_b[b];
          match_of_a[other] = -1;
          if (augment((uint32_t)other)) {
            match_of_b[b] = (int)a;
            match_of_a[a] = (int)b;
            return true;
          }
          match_of_a[other] = (int)b;
        }
        return false;
      }
    } M;
    std::vector<uint32_t> avail(kBanks, 0);
    for (uint32_t a = 0; a < kBanks; ++a)
      for (uint32_t b = 0; b < kBanks; ++b)
        if (left[a * kBanks + b]) avail[a] |= 1u << b;
    M.left = left.data();
    M.avail = avail.data();
    { const char* e = getenv("ZKI_MATCH_CHOICES"); M.choices = e ? (uint32_t)atoi(e) : 1; }
    int* const match_of_b = M.match_of_b;
    int* const match_of_a = M.match_of_a;
    auto alive = [&](uint32_t a, uint32_t b) {   // lazy deletion of the placed ops at the back of the edge's list
      const uint32_t c = a * kBanks + b;
      while (cell_end[c] > cell_begin[c] && used[cell_item[cell_end[c] - 1] & ~kFlip]) --cell_end[c];
      return cell_end[c] > cell_begin[c];
    };
    const bool shared_first = offset % 64 != 0;
    Carry carry;
    for (size_t gi = 0; gi < n_groups; ++gi) {
      const size_t g = group_at(gi, n_groups, shared_first);
      if (shared_first && g == 0) carry = wait_carry();   // the banks the run before this one uses in the block the two share
      uint32_t lanes = 0;
      for (uint32_t lane = 0; lane < kBanks; ++lane) lanes += pos_of(g, lane) != (size_t)kInf;
      if (!lanes) continue;
      for (uint32_t b = 0; b < kBanks; ++b) match_of_b[b] = match_of_a[b] = -1;
      // the first block may be shared with the run before this one: its banks are taken
      const bool shared = g < 2 && shared_first;
      const uint32_t blocked_a = shared ? carry.a[g % 2] : 0;
      M.taken_b = shared ? carry.b[g % 2] : 0;
      M.rotation = (uint32_t)(g * 11);
      uint32_t matched = 0;
      for (uint32_t t = 0; t < kBanks && matched < lanes; ++t) {
        const uint32_t a = (t + (uint32_t)g) % kBanks;
        if ((blocked_a >> a) & 1 || !avail[a]) continue;
        M.visited = 0;
        if (M.augment(a)) ++matched;
      }
      uint32_t lane = 0;
      for (uint32_t a = 0; a < kBanks; ++a) {
        if (match_of_a[a] < 0) continue;
        const uint32_t b = (uint32_t)match_of_a[a];
        if (!alive(a, b)) continue;   // (cannot happen while `left` is exact; a hole is harmless)
        const uint32_t e = cell_item[cell_end[a * kBanks + b] - 1];
        const uint32_t r = e & ~kFlip;
        used[r] = 1;
        if (--left[a * kBanks + b] == 0) avail[a] &= ~(1u << b);
        if (a != b && --left[b * kBanks + a] == 0) avail[b] &= ~(1u << a);
        if (e & kFlip) {
          flips.push_back(run[r]);
          bank_xy[r] = (uint16_t)((bank_xy[r] % kBanks) * kBanks + bank_xy[r] / kBanks);
        }
        while (lane < kBanks && (pos_of(g, lane) == (size_t)kInf || out[pos_of(g, lane)] != kInf)) ++lane;
        out[pos_of(g, lane)] = r;
      }
    }
    // What no matching could take (the last tenth of a run): a group that cannot be conflict-free should at least spread
    // its ops over the banks.  Hole by hole: the leftover op (either way round) whose operand banks the group uses least
    // so far -- as they come, these groups hit a bank as often as random ones do (3.4 times; C4: 13 % of the LDS cycles).
    std::vector<uint32_t> left_ops;
    for (size_t r = 0; r < cnt; ++r)
      if (!used[r]) left_ops.push_back((uint32_t)r);
    if (getenv("ZKI_MATCH_DEBUG")) fprintf(stderr, "[match] run of %zu ops at offset %llu: %zu left over\n", cnt, (unsigned long long)offset, left_ops.size());
    if (!left_ops.empty()) {
      std::vector<uint8_t> cnt_a(n_groups * kBanks, 0), cnt_b(n_groups * kBanks, 0);
      auto bank_a = [&](uint32_t r) { return (uint32_t)bank_xy[r] / kBanks; };   // (of the op at run position r, as it stands now)
      auto bank_b = [&](uint32_t r) { return (uint32_t)bank_xy[r] % kBanks; };
      for (size_t g = 0; g < n_groups; ++g) {
        if (g < 2 && shared_first)
          for (uint32_t bk = 0; bk < kBanks; ++bk) {
            cnt_a[g * kBanks + bk] = (carry.a[g % 2] >> bk) & 1;
            cnt_b[g * kBanks + bk] = (carry.b[g % 2] >> bk) & 1;
          }
        for (uint32_t lane = 0; lane < kBanks; ++lane) {
          const size_t q = pos_of(g, lane);
          if (q == (size_t)kInf || out[q] == kInf) continue;
          ++cnt_a[g * kBanks + bank_a(out[q])];
          ++cnt_b[g * kBanks + bank_b(out[q])];
        }
      }
      for (size_t g = 0; g < n_groups; ++g)
        for (uint32_t lane = 0; lane < kBanks; ++lane) {
          const size_t q = pos_of(g, lane);
          if (q == (size_t)kInf || out[q] != kInf) continue;
          size_t best = 0;
          uint32_t best_cost = ~0u;
          bool best_flip = false;
          for (size_t c = 0; c < left_ops.size() && best_cost; ++c) {
            const uint32_t x = bank_a(left_ops[c]), y = bank_b(left_ops[c]);
            // (multiplicity the op would see on its two banks: the heavier one counts most)
            auto cost = [&](uint32_t a, uint32_t b) {
              const uint32_t ca = cnt_a[g * kBanks + a], cb = cnt_b[g * kBanks + b];
              return std::max(ca, cb) * 64u + ca + cb;
            };
            const uint32_t straight = cost(x, y), flipped = cost(y, x);
            if (straight < best_cost) { best_cost = straight; best = c; best_flip = false; }
            if (flipped < best_cost) { best_cost = flipped; best = c; best_flip = true; }
          }
          const uint32_t r = left_ops[best];
          left_ops[best] = left_ops.back();
          left_ops.pop_back();
          if (best_flip) {
            flips.push_back(run[r]);
            bank_xy[r] = (uint16_t)((bank_xy[r] % kBanks) * kBanks + bank_xy[r] / kBanks);
          }
          out[q] = r;
          ++cnt_a[g * kBanks + bank_a(r)];
          ++cnt_b[g * kBanks + bank_b(r)];
        }
    }
    for (size_t q = 0; q < cnt; ++q) order[k0 + q] = run[out[q]];
    // the banks the run uses in the block its successor will share with it (what note_run computes from ra / rb)
    const uint64_t end = offset + cnt;
    if (end % 64 == 0) {
      carry_out = Carry();
    } else {
      const uint64_t last_block = (end - 1) / 64;
      carry_out = (offset / 64 != last_block || offset % 64 == 0) ? Carry() : carry;   // (else: the run lies inside the shared block)
      for (uint64_t q = std::max(offset, last_block * 64); q < end; ++q) {
        const uint32_t r = out[q - offset];
        carry_out.a[q % 2] |= 1u << (bank_xy[r] / kBanks);
        carry_out.b[q % 2] |= 1u << (bank_xy[r] % kBanks);
      }
    }
  };
  auto place = [&](uint32_t i, uint32_t bank = kInf) {
    const uint32_t wl = s.level_of[i] - base;
    if (wl < strand_end.size() && strand_end[wl] && closed(i) && std::max(last_use[i], s.level_of[i]) - base < strand_end[wl] &&
        (!lds_free.empty() || lds_next < lds_cap)) {
      // strand-local: an LDS slot (released like any other, at the level of its last reader)
      uint32_t k;
      if (!lds_free.empty()) {
        k = lds_free.back();
        lds_free.pop_back();
      } else {
        k = lds_next++;
        strand_lds_slots[strand_first] = lds_next;
      }
      s.slot_of[i] = kSlotInLds | k;
    } else {
      s.slot_of[i] = take_slot(bank);
    }
    if (opt.retain_all) return;
    if (closed(i)) {
      const uint32_t lu = std::max(last_use[i], s.level_of[i]) - base;
      expire_next[i - lo] = expire_head[lu];
      expire_head[lu] = i;
    } else {
      open_list.push_back(i);
    }
  };
  auto gather_level = [&](uint32_t l) {
    for (uint32_t h = expire_head[l]; h != kInf; h = expire_next[h - lo]) batch.push_back(s.slot_of[h]);
    for (uint32_t e = ext_head[l]; e != kInf; e = ext[e].next) batch.push_back(s.slot_of[ext[e].h]);
  };
  auto release_level = [&](uint32_t l) {
    gather_level(l);
    release_batch();
  };
  auto row_class = [&](uint8_t k) { return k == TK_AND || k == TK_XOR || k == TK_NOT || k == TK_COPY; };
  for (uint32_t l = 0; l < n_wlevels; ++l) {
    if (!opt.retain_all && l > 0) release_level(l - 1);
    if (strand_end[l] && (l == 0 || strand_end[l - 1] != strand_end[l])) {   // a strand begins: its LDS is its own
      lds_free.clear();
      lds_next = 0;
      strand_first = l;
    }
    // GF(2): a level wide enough for a launch of its own runs as rows of the LDS-resident kernel (lds_program.cpp), two
    // ops per thread in the order of its row SEQUENCE -- the and, xor, not and copy ops of the level, in that order
    // (order_by_level); a narrower one joins a sequential segment (emit_launches)
    const bool rows_level = s.boolean_path && level_start[l + 1] - level_start[l] >= opt.bool_narrow_width;
    uint64_t rows0 = level_start[l + 1];   // first op of the row sequence
    for (uint64_t k = level_start[l]; k < level_start[l + 1] && rows0 == level_start[l + 1]; ++k)
      if (row_class(kind[order[k] - lo])) rows0 = k;
    const bool wide_rows = level_start[l + 1] - rows0 >= 2 * kBanks;   // long enough for whole groups of 32 lanes
    if (s.boolean_path && rows0 < level_start[l + 1]) {
      // the (level, kind) runs of the row sequence, ordered for the LDS banks before any of their ops gets a slot (the order
      // only looks at operands, i.e. at earlier levels) -- concurrently, each run waiting for its predecessor's final order
      // only to fill its part of the 64-position block the two share
      struct Run {
        uint64_t k0, k1;
        bool ordered;
        Carry carry;                 // in: what the runs before left in the shared block; out (note_run): what this one leaves
        std::vector<uint32_t> flips; // ops whose operands are to be swapped (applied behind the parallel section)
        std::atomic<bool> done{false};
      };
      std::deque<Run> runs;
      for (uint64_t k = rows0; k < level_start[l + 1];) {
        uint64_t e = k;
        while (e < level_start[l + 1] && kind[order[e] - lo] == kind[order[k] - lo]) ++e;
        runs.emplace_back();
        runs.back().k0 = k;
        runs.back().k1 = e;
        runs.back().ordered = banked && rows_level && e - k >= 2 * kBanks && n_inputs(kind[order[k] - lo]) >= 1;
        k = e;
      }
      const auto tq = std::chrono::steady_clock::now();
      auto do_run = [&](size_t r) {
        Run& R = runs[r];
        // the carry a run starts from: its predecessor's, final once that run is done
        auto wait_carry = [&]() -> const Carry& {
          if (r > 0) {
            while (!runs[r - 1].done.load(std::memory_order_acquire)) std::this_thread::yield();
            R.carry = runs[r - 1].carry;
          }
          return R.carry;
        };
        const std::function<const Carry&()> wc = wait_carry;
        const auto tr = std::chrono::steady_clock::now();
        if (R.ordered && n_inputs(kind[order[R.k0] - lo]) == 2) {
          Carry out;
          bank_order_two(R.k0, R.k1, R.k0 - rows0, wc, R.flips, out);
          wait_carry();   // (a run that does not start inside a shared block never asked)
          R.carry = out;
          R.done.store(true, std::memory_order_release);
          dbg_run_ns[std::min<size_t>(r, 3)] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tr).count();
          return;
        }
        if (R.ordered) bank_order(R.k0, R.k1, R.k0 - rows0, wc);
        dbg_run_ns[std::min<size_t>(r, 3)] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tr).count();
        wait_carry();
        note_run(R.k0, R.k1, R.k0 - rows0, R.carry);
        R.done.store(true, std::memory_order_release);
      };
      size_t n_ordered = 0;
      for (const Run& R : runs) n_ordered += R.ordered;
      if (n_ordered >= 2 && threads > 1) {
        // (tasks are claimed in run order, so the run a task waits for is always under way)
        const std::function<void(uint32_t)> fn = [&](uint32_t r) { do_run(r); };
        pool().run((uint32_t)runs.size(), fn);
      } else {
        for (size_t r = 0; r < runs.size(); ++r) do_run(r);
      }
      for (const Run& R : runs)
        for (uint32_t i : R.flips) std::swap(ra[i - lo], rb[i - lo]);
      t_bank_order += std::chrono::duration<double>(std::chrono::steady_clock::now() - tq).count();
    }
    for (uint64_t k = level_start[l]; k < level_start[l + 1]; ++k) {
      const uint32_t i = order[k];
      if (kind[i - lo] == TK_ASSERT || kind[i - lo] == TK_NOP) continue;
      // position of the op in the row sequence (the thread that executes it is position / 2), for the ops that run as rows
      prefer_bank = wide_rows;
      place(i, (rows_level && row_class(kind[i - lo])) ? (uint32_t)(k - rows0) : kInf);
      if (!pair_second.empty() && pair_second[i - lo] != kInf) place(pair_second[i - lo]);  // the second value of a pair entry
    }
  }
  // what expires at the window's last level is free for the next window
  // (GF(2): kept for the batch the next window opens with, so that it is one batch as in a one-window schedule)
  if (!opt.retain_all && n_wlevels) {
    gather_level(n_wlevels - 1);
    if (s.boolean_path) pending_release.swap(batch);
    else release_batch();
  }
}

void StreamScheduler::Impl::order_levels() {
  // ---- order of the entries inside a level (independent per level: done on `threads` threads) ----
  if (!opt.sort_by_operand) return;
  // GF(2): the order inside a run is final when the slots are assigned -- the two results of a thread (run positions 2t,
  // 2t + 1) are the halves of one slot pair, and a bank-aware schedule has ordered the run for the LDS banks by then
  if (s.boolean_path) return;
  parallel_levels(n_wlevels, [&](uint32_t l) {
    // inside a (level, kind) run, order the ops by the slot of their first operand: gates that read
    // the same wire become neighbours (same workgroup), so the repeat read is an L1/L2 hit
    uint64_t k = level_start[l];
    while (k < level_start[l + 1]) {
      uint64_t e = k;
      const uint8_t kd = kind[order[k] - lo];
      while (e < level_start[l + 1] && kind[order[e] - lo] == kd) ++e;
      if (n_inputs(kd) >= 1 && e - k > 1 && (opt.sort_by_operand != 3 || s.boolean_path))
        std::stable_sort(order.begin() + k, order.begin() + e, [&](uint32_t x, uint32_t y) {
          const uint32_t ax = inner(ra[x - lo]) ? ra[ra[x - lo] - lo] : ra[x - lo];
          const uint32_t ay = inner(ra[y - lo]) ? ra[ra[y - lo] - lo] : ra[y - lo];
          return s.slot_of[ax] < s.slot_of[ay];
        });
      k = e;
    }
    if (opt.sort_by_operand >= 2 && !s.boolean_path) {
      // the Add/Mul entries of a level come first (counting sort by kind) and run in a kernel instantiation of
      // their own (engine.hip launch_one): the shared-operand walk orders the two parts separately
      uint64_t mid = level_start[l];
      while (mid < level_start[l + 1] && (kind[order[mid] - lo] == TK_ADD || kind[order[mid] - lo] == TK_MUL)) ++mid;
      const uint64_t cut[3] = {level_start[l], mid, level_start[l + 1]};
      for (int part = 0; part < 2; ++part)
        if (cut[part + 1] - cut[part] > 8)
          locality_order(order.data() + cut[part], cut[part + 1] - cut[part],
                         [&](uint32_t i, uint32_t* out) { return gathered(i, out); });
    }
  });
}

void StreamScheduler::Impl::emit_entries() {
  // ---- device ops ----------------------------------------------------------
  const size_t n_live = order.size();
  if (s.fused) {
    const size_t at = s.ops2.size();
    s.ops2.resize(at + n_live);
    parallel_slices(n_live, 1 << 16, [&](size_t k_lo, size_t k_hi) {
    for (size_t k = k_lo; k < k_hi; ++k) {
      const uint32_t i = order[k];
      const uint8_t kd = kind[i - lo];
      DevOp2 d{0, kd, 0, 0, 0, 0, 0, 0};
      d.dst = s.slot_of[i] == kNoWire ? 0 : s.slot_of[i];
      auto operand = [&](uint32_t h, uint32_t* x0, uint32_t* x1, int shift) {
        if (inner(h)) {  // evaluated inside this entry (ST_PAIR_SECOND has a slot of its own)
          *x0 = s.slot_of[ra[h - lo]];
          *x1 = s.slot_of[rb[h - lo]];
          d.kind |= (kind[h - lo] == TK_ADD ? 1u : 2u) << shift;
        } else {
          *x0 = s.slot_of[h];
        }
      };
      const uint32_t x = ra[i - lo], y = rb[i - lo];
      switch (kd) {
        case TK_ADD: case TK_MUL:
          if (!pair_second.empty() && pair_second[i - lo] != kInf) {
            // pair entry: the shared producer goes first (Add/Mul commute), then the second gate's result slot
            // and its own operand
            const uint32_t j = pair_second[i - lo];
            const bool shared_is_a = x >= lo && st(x) == ST_PAIR_SHARED;
            const uint32_t sh = shared_is_a ? x : y, other = shared_is_a ? y : x;
            operand(sh, &d.a0, &d.a1, 8);
            operand(other, &d.b0, &d.b1, 10);
            d.kind |= (kind[j - lo] == TK_ADD ? 1u : 2u) << 12;
            d.pad0 = s.slot_of[j];
            d.pad1 = s.slot_of[ra[j - lo] == sh ? rb[j - lo] : ra[j - lo]];
          } else {
            operand(x, &d.a0, &d.a1, 8);
            operand(y, &d.b0, &d.b1, 10);
          }
          break;
        case TK_AND: case TK_XOR:   // (an operand that is a copy of an input: the input itself, bit_operand)
          d.a0 = opnd_code[0][i - lo] ? (kOperandIsSource | opnd_code[0][i - lo]) : s.slot_of[x];
          d.b0 = opnd_code[1][i - lo] ? (kOperandIsSource | opnd_code[1][i - lo]) : s.slot_of[y];
          break;
        case TK_ADDC: case TK_MULC:
          d.a0 = s.slot_of[x];
          d.b0 = y;
          break;
        case TK_COPY: case TK_NZ: d.a0 = s.slot_of[x]; break;
        case TK_NOT: d.a0 = s.slot_of[x]; d.a1 = sink_code[i - lo]; break;   // a1: the unreduced source behind the operand
        case TK_CONST: d.a0 = x; break;
        case TK_INSTANCE: case TK_WITNESS: case TK_CARRY: d.a0 = x; break;
        case TK_ASSERT:
          d.a0 = s.slot_of[x];
          d.b0 = y;
          d.a1 = sink_code[i - lo];
          break;
        default: break;
      }
      s.ops2[at + k] = d;
    }
    });
    return;
  }
  const size_t at = s.ops.size();
  s.ops.resize(at + n_live);
  parallel_slices(n_live, 1 << 16, [&](size_t k_lo, size_t k_hi) {
  for (size_t k = k_lo; k < k_hi; ++k) {
    const uint32_t i = order[k];
    const uint8_t kd = kind[i - lo];
    DevOp d{0, 0, 0, kd};
    d.dst = s.slot_of[i] == kNoWire ? 0 : s.slot_of[i];
    const uint32_t x = ra[i - lo], y = rb[i - lo];
    switch (kd) {
      case TK_ADD: case TK_MUL:
        d.a = s.slot_of[x];
        d.b = s.slot_of[y];
        break;
      case TK_AND: case TK_XOR:   // (an operand that is a copy of an input: the input itself, bit_operand)
        d.a = opnd_code[0][i - lo] ? (kOperandIsSource | opnd_code[0][i - lo]) : s.slot_of[x];
        d.b = opnd_code[1][i - lo] ? (kOperandIsSource | opnd_code[1][i - lo]) : s.slot_of[y];
        break;
      case TK_ADDC: case TK_MULC:
        d.a = s.slot_of[x];
        d.b = y;
        break;
      case TK_COPY: case TK_NZ: d.a = s.slot_of[x]; break;
      case TK_NOT: d.a = s.slot_of[x]; d.b = field.is_two ? 0 : sink_code[i - lo]; break;   // b: the unreduced source behind the operand
      case TK_CONST:
        // GF(2): a constant >= 2 that only zero tests read (through copies) is `non-zero` to them: the constant 1
        d.a = (field.is_two && src_pos[i] && src_zero_test[i] && !src_other[i]) ? kSyntheticOne : x;
        break;
      case TK_INSTANCE: case TK_WITNESS: case TK_CARRY: d.a = x; break;
      case TK_ASSERT:
        d.a = s.slot_of[x];
        d.b = y;
        d.dst = field.is_two ? 0 : sink_code[i - lo];
        break;
      default: break;
    }
    s.ops[at + k] = d;
  }
  });
}

void StreamScheduler::Impl::emit_launches() {
  // ---- launches -------------------------------------------------------------
  const uint64_t at = (s.fused ? s.ops2.size() : s.ops.size()) - order.size();
  uint32_t l = 0;
  while (l < n_wlevels) {
    const uint64_t width = level_start[l + 1] - level_start[l];
    s.max_level_width = std::max<uint32_t>(s.max_level_width, (uint32_t)width);
    Launch L;
    L.first = (uint32_t)(at + level_start[l]);
    L.level_begin = base + l;
    L.window = n_windows;
    const uint64_t k0 = level_start[l];
    uint64_t k1;
    // a level too narrow to pay for a launch of its own goes into a sequential launch with its narrow neighbours: a
    // strand (workgroup per lane block, barrier between levels) in the fused format, one wave per lane block otherwise
    const uint32_t narrow = s.fused ? std::max(opt.strand_width, opt.narrow_width) : s.boolean_path ? opt.bool_narrow_width : opt.narrow_width;
    if (width >= narrow) {
      L.count = (uint32_t)width;
      L.ops_per_wave = 1;
      L.level_end = base + l + 1;
      k1 = level_start[l + 1];
      ++l;
    } else {
      uint32_t e = l;
      while (e < n_wlevels && level_start[e + 1] - level_start[e] < narrow) ++e;
      L.count = (uint32_t)(level_start[e] - level_start[l]);
      // the level bounds of the strand: one interval per NON-EMPTY level (a level whose gates were all absorbed by their
      // readers has nothing to run and needs no barrier)
      L.level_ptr = (uint32_t)s.strand_level_ptr.size();
      s.strand_level_ptr.push_back(0);
      for (uint32_t q = l; q < e; ++q)
        if (level_start[q + 1] > level_start[q]) {
          s.strand_level_ptr.push_back((uint32_t)(level_start[q + 1] - level_start[l]));
          ++L.strand_levels;
        }
      L.lds_slots = l < strand_lds_slots.size() ? strand_lds_slots[l] : 0;
      L.ops_per_wave = std::max<uint32_t>(L.count, 1);
      L.sequential = true;
      L.level_end = base + e;
      k1 = level_start[e];
      l = e;
    }
    for (uint64_t k = k0; k < k1; ++k) {
      const uint8_t kd = kind[order[k] - lo];
      if (!L.sequential && k == k0 + L.hot_count && (kd == TK_ADD || kd == TK_MUL)) ++L.hot_count;
      if (!s.boolean_path && (kd == TK_AND || kd == TK_XOR)) L.has_bitops = true;
    }
    s.has_bitops = s.has_bitops || L.has_bitops;
    if (L.count) s.launches.push_back(L);
  }
}

// Strands, last step: what a strand reads out of the WIRE TABLE -- values made before it, e.g. the Switch weights of a loop
// body, computed level-wide in front of the chain -- costs the entry that reads it a trip to L2 or HBM in the middle of the
// dependency chain (measured: ~1,300 cycles per iteration of the chained structured relation, profiles/r04_strand_stamps.txt).
// The strand has waves to spare, so such an operand is COPIED into LDS one to three levels earlier by an entry of its own
// (kind copy, wire table -> kSlotInLds) on a spare wave of a level that has a product to hide it behind, and the reader is
// pointed at the LDS value.  Entries only: tape handles, levels and the slot allocation are what they were; the copies use
// LDS values above the strand's own, recycled once their reader has run.
void StreamScheduler::Impl::prefetch_strand_inputs(size_t first_launch) {
  if (!s.fused || !opt.strand_lds || !(opt.strand_prefetch || opt.strand_merge || opt.strand_reassociate || opt.strand_split_inputs)) return;
  const uint32_t value_bytes = ((field.nwords + 3) / 4) * 64 * 16;
  const uint32_t lds_cap = std::min<uint32_t>(1024, kStrandLdsBytes / std::max<uint32_t>(value_bytes, 1));
  int64_t shift = 0;   // entries inserted in front of the launch being looked at
  for (size_t li = first_launch; li < s.launches.size(); ++li) {
    Launch& L = s.launches[li];
    L.first = (uint32_t)(L.first + shift);
    if (!L.sequential || L.strand_levels < 2) continue;
    const uint32_t nl = L.strand_levels;
    std::vector<uint32_t> lp(s.strand_level_ptr.begin() + L.level_ptr, s.strand_level_ptr.begin() + L.level_ptr + nl + 1);
    std::vector<DevOp2> ent(s.ops2.begin() + L.first, s.ops2.begin() + L.first + L.count);
    // slot operands of an entry (pointers into it), by kind
    auto operands = [](DevOp2& d, uint32_t* out[5]) {
      const uint32_t k = d.kind & 0xFF, ea = (d.kind >> 8) & 3, eb = (d.kind >> 10) & 3, pair = (d.kind >> 12) & 3;
      int n = 0;
      switch (k) {
        case TK_ADD: case TK_MUL:
          out[n++] = &d.a0;
          if (ea) out[n++] = &d.a1;
          out[n++] = &d.b0;
          if (eb) out[n++] = &d.b1;
          if (pair) out[n++] = &d.pad1;
          break;
        case TK_ADDC: case TK_MULC: case TK_COPY: case TK_NZ: case TK_NOT: case TK_ASSERT: case TK_INPUT_CONV: out[n++] = &d.a0; break;
        case TK_AND: case TK_XOR:
          if (!(d.a0 & kOperandIsSource)) out[n++] = &d.a0;
          if (!(d.b0 & kOperandIsSource)) out[n++] = &d.b0;
          break;
        default: break;
      }
      return n;
    };
    auto writes = [](const DevOp2& d, uint32_t out[2]) {
      const uint32_t k = d.kind & 0xFF;
      int n = 0;
      if (k != TK_ASSERT && k != TK_NOP) out[n++] = d.dst;
      if ((k == TK_ADD || k == TK_MUL) && ((d.kind >> 12) & 3)) out[n++] = d.pad0;
      return n;
    };
    auto heavy_entry = [](const DevOp2& d) {
      const uint32_t k = d.kind & 0xFF;
      return k == TK_MUL || k == TK_MULC || k == TK_INSTANCE || k == TK_WITNESS || k == TK_CARRY || k == TK_INPUT_CONV ||
             (k == TK_ADD && (((d.kind >> 8) & 3) == 2 || ((d.kind >> 10) & 3) == 2 || ((d.kind >> 12) & 3) == 2));
    };
    // Products re-associated off the dependency chain.  A fused entry (z * x) * y -- or (x * y) * z -- whose z was made in the
    // level right in front of it while x and y have been there for two levels or more (a witness, a Switch weight) is two
    // dependent products on the chain; x * y does not depend on the chain at all: it becomes an entry of its own on a spare
    // wave of an earlier level (an LDS value above the strand's own) and the chain keeps z * (x * y), ONE product.  Field
    // multiplication is associative and every value canonical, so the result is the same bits.
    uint32_t reassoc_lds = 0;   // LDS values the products taken off the chain live in
    size_t n_reassoc = 0, n_split = 0;
    if ((opt.strand_reassociate || opt.strand_split_inputs) && L.lds_slots < lds_cap) {
      const uint32_t cap = std::min<uint32_t>(16, lds_cap - L.lds_slots);
      struct Slot { uint32_t loaded = 0, needed = 0; };
      std::vector<Slot> rs;
      std::unordered_map<uint32_t, uint32_t> last_write;   // slot -> level of its latest write (levels in front of q)
      std::vector<std::vector<DevOp2>> add(nl);
      std::vector<uint32_t> cnt(nl, 0);
      for (uint32_t q = 0; q < nl; ++q) cnt[q] = lp[q + 1] - lp[q];
      auto avail = [&](uint32_t slot) -> int64_t {   // -1: there before the strand
        auto it = last_write.find(slot);
        if (it == last_write.end()) return (slot & kSlotInLds) ? INT64_MAX : -1;   // (an LDS value nobody wrote: leave it alone)
        return it->second;
      };
      auto lds_value = [&](uint32_t t, uint32_t q) -> uint32_t {   // an LDS value written at level t, last read at level q
        uint32_t r = kInf;
        for (uint32_t j = 0; j < rs.size() && r == kInf; ++j)
          if (rs[j].needed < t) r = j;
        if (r == kInf) {
          if (rs.size() >= cap) return kInf;
          rs.emplace_back();
          r = (uint32_t)rs.size() - 1;
        }
        rs[r].loaded = t;
        rs[r].needed = q;
        reassoc_lds = std::max(reassoc_lds, r + 1);
        return kSlotInLds | (L.lds_slots + r);
      };
      for (uint32_t q = 0; q < nl; ++q) {
        // An input of the strand -- fetch from HBM, canonical check, conversion (a product) -- is the longest entry a level
        // of a chain can hold (3,600 cycles on the chained structured relation, beside two products of 1,300).  It is taken
        // apart: the words as they lie in the buffer go into an LDS value on a spare wave of an earlier level (the one with
        // the most room of the three in front), the conversion stays where the input was.
        for (uint32_t k = lp[q]; k < lp[q + 1] && q >= 1 && opt.strand_split_inputs; ++k) {
          DevOp2& e = ent[k];
          const uint32_t kind = e.kind & 0xFF;
          if (kind != TK_INSTANCE && kind != TK_WITNESS) continue;
          uint32_t t = kInf, room = 0;
          for (uint32_t back = 1; back <= 3 && back <= q; ++back) {
            const uint32_t used = cnt[q - back] + (uint32_t)add[q - back].size();
            if (used < 4 && 4 - used > room) { t = q - back; room = 4 - used; }
          }
          if (t == kInf) continue;
          const uint32_t m = lds_value(t, q);
          if (m == kInf) continue;
          const uint32_t stream = kind == TK_WITNESS ? 1u : 0u, position = e.a0;
          add[t].push_back(DevOp2{m, TK_INPUT_RAW, position, stream, 0, 0, 0, 0});
          e.kind = TK_INPUT_CONV;
          e.a0 = m;
          e.a1 = stream;
          e.b0 = position;
          ++n_split;
        }
        for (uint32_t k = lp[q]; k < lp[q + 1] && q >= 2 && opt.strand_reassociate; ++k) {
          DevOp2& e = ent[k];
          const uint32_t kind = e.kind & 0xFF, ea = (e.kind >> 8) & 3, eb = (e.kind >> 10) & 3, pair = (e.kind >> 12) & 3;
          if (kind != TK_MUL || pair || !((ea == 2 && eb == 0) || (ea == 0 && eb == 2))) continue;
          // the three factors: i0 * i1 is the inner product, o the outer factor
          const uint32_t i0 = ea == 2 ? e.a0 : e.b0, i1 = ea == 2 ? e.a1 : e.b1, o = ea == 2 ? e.b0 : e.a0;
          const int64_t av[3] = {avail(i0), avail(i1), avail(o)};
          const int64_t late = (int64_t)q - 1;
          const int n_late = (av[0] == late) + (av[1] == late) + (av[2] == late);
          if (n_late != 1 || av[0] > late || av[1] > late || av[2] > late) continue;
          // z: the late factor; x, y: the other two
          uint32_t z, x, y;
          if (av[0] == late) { z = i0; x = i1; y = o; }
          else if (av[1] == late) { z = i1; x = i0; y = o; }
          else continue;   // the outer factor is the late one: the inner product is off the chain already
          const int64_t ready = std::max(avail(x), avail(y));
          uint32_t t = kInf;
          for (uint32_t back = 1; back <= 3 && back <= q; ++back) {
            if ((int64_t)(q - back) <= ready) break;
            if (cnt[q - back] + add[q - back].size() < 4) { t = q - back; break; }
          }
          if (t == kInf) continue;
          const uint32_t m = lds_value(t, q);
          if (m == kInf) continue;
          add[t].push_back(DevOp2{m, TK_MUL, x, 0, y, 0, 0, 0});
          e.kind = TK_MUL;
          e.a0 = z;
          e.a1 = 0;
          e.b0 = m;
          e.b1 = 0;
          ++n_reassoc;
        }
        for (uint32_t k = lp[q]; k < lp[q + 1]; ++k) {
          uint32_t w[2];
          const int nw = writes(ent[k], w);
          for (int j = 0; j < nw; ++j) last_write[w[j]] = q;
        }
      }
      if (n_reassoc + n_split) {
        std::vector<DevOp2> out;
        out.reserve(ent.size() + n_reassoc + n_split);
        std::vector<uint32_t> nlp(1, 0);
        for (uint32_t q = 0; q < nl; ++q) {
          out.insert(out.end(), ent.begin() + lp[q], ent.begin() + lp[q + 1]);
          out.insert(out.end(), add[q].begin(), add[q].end());
          nlp.push_back((uint32_t)out.size());
        }
        ent.swap(out);
        lp.swap(nlp);
      }
    }
    const uint32_t lds_base = L.lds_slots + reassoc_lds;   // the copies' LDS values come behind
    std::vector<uint8_t> written(std::max(n_slots, s.n_slots) + 1, 0);   // wire-table slots the strand itself writes: never prefetched
    std::vector<uint32_t> count(nl, 0), heavy(nl, 0);
    for (uint32_t q = 0; q < nl; ++q)
      for (uint32_t k = lp[q]; k < lp[q + 1]; ++k) {
        uint32_t w[2];
        const int nw = writes(ent[k], w);
        for (int j = 0; j < nw; ++j)
          if (!(w[j] & kSlotInLds) && w[j] < written.size()) written[w[j]] = 1;
        ++count[q];
        heavy[q] += heavy_entry(ent[k]);
      }
    struct Ring {
      uint32_t slot = kInf;       // wire-table slot it holds
      uint32_t loaded = 0, needed = 0;   // level of the copy, level of its last reader so far
    };
    std::vector<Ring> ring;
    const uint32_t ring_cap = opt.strand_prefetch && lds_base < lds_cap ? std::min<uint32_t>(32, lds_cap - lds_base) : 0;
    std::vector<std::vector<DevOp2>> extra(nl);
    uint32_t ring_used = 0;
    for (uint32_t q = 1; q < nl && ring_cap; ++q)
      for (uint32_t k = lp[q]; k < lp[q + 1]; ++k) {
        uint32_t* ops[5];
        const int n = operands(ent[k], ops);
        for (int j = 0; j < n; ++j) {
          const uint32_t slot = *ops[j];
          if ((slot & kSlotInLds) || slot >= written.size() || written[slot]) continue;
          // in LDS already (an earlier reader of the strand brought it)?
          uint32_t r = kInf;
          for (uint32_t x = 0; x < ring.size(); ++x)
            if (ring[x].slot == slot && ring[x].loaded < q) r = x;
          if (r == kInf) {
            // a level for the copy: the nearest of the three in front of the reader that has a product and a wave to spare
            uint32_t t = kInf;
            for (uint32_t back = 1; back <= 3 && back <= q; ++back)
              if (heavy[q - back] && count[q - back] + extra[q - back].size() < 4) { t = q - back; break; }
            if (t == kInf) continue;
            // a ring value whose reader has run before level t (its LDS value may be overwritten at t)
            for (uint32_t x = 0; x < ring.size() && r == kInf; ++x)
              if (ring[x].needed < t) r = x;
            if (r == kInf) {
              if (ring.size() >= ring_cap) continue;
              ring.emplace_back();
              r = (uint32_t)ring.size() - 1;
            }
            ring[r].slot = slot;
            ring[r].loaded = t;
            ring[r].needed = q;
            ring_used = std::max(ring_used, r + 1);
            DevOp2 c{kSlotInLds | (lds_base + r), TK_COPY, slot, 0, 0, 0, 0, 0};
            extra[t].push_back(c);
          }
          ring[r].needed = std::max(ring[r].needed, q);
          *ops[j] = kSlotInLds | (lds_base + r);
        }
      }
    size_t n_extra = 0;
    for (const auto& e : extra) n_extra += e.size();
    if (getenv("ZKI_SCHED_PROFILE"))
      fprintf(stderr, "[schedule] strand of %u levels, %u entries, %u LDS values: %zu wire-table operands copied ahead (ring %u of %u)\n",
              nl, L.count, L.lds_slots, n_extra, ring_used, ring_cap);
    // the launch's entries with the copies behind the entries of their level, and its level bounds
    std::vector<DevOp2> out;
    out.reserve(ent.size() + n_extra);
    std::vector<uint32_t> nlp(1, 0);
    for (uint32_t q = 0; q < nl; ++q) {
      out.insert(out.end(), ent.begin() + lp[q], ent.begin() + lp[q + 1]);
      out.insert(out.end(), extra[q].begin(), extra[q].end());
      nlp.push_back((uint32_t)out.size());
    }
    // Levels that need no barrier between them.  Entry i of a level runs on wave i % 4, a wave its entries in order, and
    // a wave's LDS accesses happen in order: level B can join the level A in front of it (one barrier and one wait for the
    // program entry less per level joined -- ~400 of the ~1,300 cycles a level of the chained structured relation costs
    // beyond its arithmetic) when (i) no entry of B touches what A writes or writes what A reads, or (ii) ONE entry of B
    // does, only entries of A that run on wave 0, and only through LDS values: that entry then follows them on wave 0
    // (the next position that is a multiple of 4, the gap filled with B's other entries or with no-ops), the others run
    // beside them as they would have.  A chain of single entries becomes one level that wave 0 walks alone.
    size_t n_merged = 0, n_nops = 0;
    if (opt.strand_merge && nlp.size() > 2) {
      constexpr size_t kMaxJoined = 64;   // entries of a joined level
      auto in = [](const std::vector<uint32_t>& v, uint32_t x) { return std::find(v.begin(), v.end(), x) != v.end(); };
      std::vector<DevOp2> merged;
      std::vector<uint32_t> mlp(1, 0);
      merged.reserve(out.size() + 16);
      std::vector<DevOp2> cur(out.begin() + nlp[0], out.begin() + nlp[1]);
      const DevOp2 nop{0, TK_NOP, 0, 0, 0, 0, 0, 0};
      for (size_t q = 1; q + 1 < nlp.size(); ++q) {
        std::vector<DevOp2> B(out.begin() + nlp[q], out.begin() + nlp[q + 1]);
        // what the entries of cur read and write, those of wave 0 and those of the other waves apart
        std::vector<uint32_t> wr[2], rd[2];
        for (size_t k = 0; k < cur.size(); ++k) {
          uint32_t* ops[5];
          const int n = operands(cur[k], ops);
          for (int j = 0; j < n; ++j) rd[k % 4 != 0].push_back(*ops[j]);
          uint32_t w[2];
          const int nw = writes(cur[k], w);
          for (int j = 0; j < nw; ++j) wr[k % 4 != 0].push_back(w[j]);
        }
        std::vector<size_t> dep;
        bool joinable = cur.size() + B.size() + 3 <= kMaxJoined;
        for (size_t k = 0; k < B.size() && joinable; ++k) {
          bool hit = false;
          uint32_t* ops[5];
          const int n = operands(B[k], ops);
          uint32_t w[2];
          const int nw = writes(B[k], w);
          for (int side = 0; side < 2; ++side) {
            for (int j = 0; j < n; ++j)
              if (in(wr[side], *ops[j])) { hit = true; joinable &= side == 0 && (*ops[j] & kSlotInLds); }
            for (int j = 0; j < nw; ++j)
              if (in(wr[side], w[j]) || in(rd[side], w[j])) { hit = true; joinable &= side == 0 && (w[j] & kSlotInLds); }
          }
          if (hit) dep.push_back(k);
        }
        joinable &= dep.size() <= 1;
        if (joinable) {
          size_t next = 0;
          auto skip_dep = [&]() { while (!dep.empty() && next == dep[0]) ++next; };
          if (!dep.empty()) {
            for (skip_dep(); cur.size() % 4 != 0; skip_dep()) {
              if (next < B.size()) cur.push_back(B[next++]);
              else { cur.push_back(nop); ++n_nops; }
            }
            cur.push_back(B[dep[0]]);
          }
          for (skip_dep(); next < B.size(); skip_dep()) cur.push_back(B[next++]);
          ++n_merged;
        } else {
          merged.insert(merged.end(), cur.begin(), cur.end());
          mlp.push_back((uint32_t)merged.size());
          cur.swap(B);
        }
      }
      merged.insert(merged.end(), cur.begin(), cur.end());
      mlp.push_back((uint32_t)merged.size());
      if (n_merged) {
        out.swap(merged);
        nlp.swap(mlp);
      }
    }
    if (getenv("ZKI_SCHED_PROFILE") && n_merged)
      fprintf(stderr, "[schedule]   %zu of its levels joined the level in front of them (%zu no-op entries)\n", n_merged, n_nops);
    if (getenv("ZKI_SCHED_PROFILE") && n_reassoc)
      fprintf(stderr, "[schedule]   %zu products taken off the dependency chain (x * y of (z * x) * y)\n", n_reassoc);
    if (getenv("ZKI_SCHED_PROFILE") && n_split) fprintf(stderr, "[schedule]   %zu inputs fetched a level or more ahead of their conversion\n", n_split);
    if (!n_extra && !n_merged && !n_reassoc && !n_split) continue;   // nothing changed
    const uint32_t old_count = L.count;
    s.ops2.erase(s.ops2.begin() + L.first, s.ops2.begin() + L.first + L.count);
    s.ops2.insert(s.ops2.begin() + L.first, out.begin(), out.end());
    std::copy(nlp.begin(), nlp.end(), s.strand_level_ptr.begin() + L.level_ptr);
    L.count = (uint32_t)out.size();
    L.ops_per_wave = std::max<uint32_t>(L.count, 1);
    L.lds_slots += reassoc_lds + ring_used;
    L.strand_levels = (uint32_t)nlp.size() - 1;
    s.n_strand_prefetches += n_extra;
    s.n_strand_levels_joined += n_merged;
    s.n_strand_reassociated += n_reassoc;
    s.n_strand_inputs_split += n_split;
    shift += (int64_t)out.size() - (int64_t)old_count;
  }
}

StreamScheduler::StreamScheduler(const FieldHost& field, const ScheduleOptions& opt) : impl_(new Impl()) {
  Impl& m = *impl_;
  m.field = field;
  m.opt = opt;
  m.threads = opt.threads ? opt.threads : std::min<uint32_t>(8, std::max(1u, std::thread::hardware_concurrency()));
  m.s.retain_all = opt.retain_all;
  m.s.boolean_path = field.is_two;
  // fused entry format whenever fusion may happen: a window cannot know whether a later one will absorb a gate
  m.s.fused = opt.fuse && !opt.retain_all && !field.is_two && !field.generic;   // (the any-modulus kernel replays unfused entries)
  m.s.window_first_op.push_back(0);
}

StreamScheduler::~StreamScheduler() { delete impl_; }

const Schedule& StreamScheduler::partial() const { return impl_->s; }

WindowResult StreamScheduler::add_window(const TapeWindow& w) {
  Impl& m = *impl_;
  WindowResult r;
  r.first_op = m.s.fused ? m.s.ops2.size() : m.s.ops.size();
  r.first_launch = (uint32_t)m.s.launches.size();
  m.lo = w.lo;
  m.hi = w.hi;
  m.final = w.final;
  m.base = m.s.n_levels;
  m.grow(w.hi);
  for (size_t k = 0; k < w.n_drops; ++k)
    if (w.drops[k] < w.hi) m.flags[w.drops[k]] |= FL_DROPPED;
  if (w.final && w.pinned)
    for (uint32_t h : *w.pinned)
      if (h < w.hi) m.flags[h] = (uint8_t)((m.flags[h] | FL_PINNED) & ~FL_DROPPED);
  const uint32_t n = w.hi - w.lo;
  static const bool profile = getenv("ZKI_SCHED_PROFILE") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto t0 = now();
  double stage[10] = {0};
  int si = 0;
  auto lap = [&] { auto t1 = now(); stage[si++] = std::chrono::duration<double>(t1 - t0).count(); t0 = t1; };
  m.track_unreduced_values(w);   // (also for an empty final window: the wires alive at the end are known only now)
  lap();
  m.n_wlevels = 0;
  m.order.clear();
  if (n) {
    // the window's own copy of the ops (operands get resolved through copy chains, kinds rewritten), written slice by slice
    m.kind.resize(n);
    m.state.resize(n);
    m.ra.resize(n);
    m.rb.resize(n);
    m.parallel_slices(n, 1 << 18, [&](size_t i0, size_t i1) {
      memcpy(m.kind.data() + i0, w.kind + i0, i1 - i0);
      memset(m.state.data() + i0, ST_ENTRY, i1 - i0);
      memcpy(m.ra.data() + i0, w.a + i0, (i1 - i0) * 4);
      memcpy(m.rb.data() + i0, w.b + i0, (i1 - i0) * 4);
      if (!m.field.is_two) return;
      // Arithmetic mod 2 on {0, 1}, lowered here and not when the pool is converted: (a + b) % 2 = xor, (a * b) % 2 = and,
      // a + c = not / copy and a * c = copy / 0 by the parity of c.  The (level, kind) runs the slots are allocated for
      // are then the runs the LDS-resident kernel's program is cut into (lds_program.cpp): its threads store their two
      // results as one aligned pair.
      for (size_t i = i0; i < i1; ++i) {
        uint8_t& k = m.kind[i];
        if (k != TK_ADD && k != TK_MUL && k != TK_ADDC && k != TK_MULC) continue;
        bool odd_const = false;
        if (k == TK_ADDC || k == TK_MULC) {
          if (w.const_parity) {
            if (m.rb[i] >= w.const_parity->size()) throw Error("scheduler: a GF(2) window names a constant its parity table does not hold");
            odd_const = (*w.const_parity)[m.rb[i]] != 0;
          } else if (w.consts && m.rb[i] < w.consts->size()) {
            odd_const = !(*w.consts)[m.rb[i]].empty() && ((*w.consts)[m.rb[i]][0] & 1);
          } else {
            throw Error("scheduler: a GF(2) window comes without its constants");
          }
        }
        if (k == TK_ADD) k = TK_XOR;
        else if (k == TK_MUL) k = TK_AND;
        else if (k == TK_ADDC) { k = odd_const ? TK_NOT : TK_COPY; m.rb[i] = 0; }
        else if (k == TK_MULC) {
          if (odd_const) { k = TK_COPY; m.rb[i] = 0; }
          else { k = TK_CONST; m.ra[i] = kSyntheticZero; m.rb[i] = 0; }
        }
      }
    });
    m.rewrite_ladders(w);
    m.propagate_copies();
    lap();
    m.levelise();
    lap();
    m.fuse_and_pair();
    m.place_strand_sources();
    lap();
    m.order_by_level();
    lap();
  } else {
    m.kind.clear();
    m.state.clear();
    m.level_start.assign(1, 0);
    m.pair_second.clear();
  }
  m.assign_slots();  // also with no ops of its own: values the last window left open may have been closed since
  if (n) {
    lap();
    m.order_levels();
    lap();
    m.emit_entries();
    m.emit_launches();
    m.prefetch_strand_inputs(r.first_launch);
    lap();
    m.s.n_levels = m.base + m.n_wlevels;
    if (profile)
      fprintf(stderr, "[schedule] window %u: %u ops, %u levels | grow+sources %.1f copies %.1f levelise %.1f fuse %.1f sort %.1f slots %.1f order %.1f emit %.1f ms\n",
              m.n_windows, n, m.n_wlevels, stage[0] * 1e3, stage[1] * 1e3, stage[2] * 1e3, stage[3] * 1e3, stage[4] * 1e3,
              stage[5] * 1e3, stage[6] * 1e3, stage[7] * 1e3);
    if (profile && m.t_bank_order > 0) fprintf(stderr, "[schedule]   of slots: %.1f ms ordering runs for the LDS banks (runs 0..3: %.1f %.1f %.1f %.1f ms)\n", m.t_bank_order * 1e3, m.dbg_run_ns[0] / 1e6, m.dbg_run_ns[1] / 1e6, m.dbg_run_ns[2] / 1e6, m.dbg_run_ns[3] / 1e6);
  }
  ++m.n_windows;
  m.s.n_slots = std::max<uint32_t>(m.n_slots, 1);
  r.n_ops = (m.s.fused ? m.s.ops2.size() : m.s.ops.size()) - r.first_op;
  r.n_launches = (uint32_t)m.s.launches.size() - r.first_launch;
  m.s.window_first_op.push_back(r.first_op + r.n_ops);
  return r;
}

Schedule StreamScheduler::finish(const std::vector<Value>& consts) {
  Impl& m = *impl_;
  Schedule& s = m.s;
  m.finish_input_modes();
  // ---- constant pool in device form -------------------------------------
  const uint32_t n_consts = (uint32_t)consts.size();
  if (s.boolean_path) {
    s.words_per_const = 1;
    s.const_words.resize(n_consts + 2);
    for (uint32_t i = 0; i < n_consts; ++i) s.const_words[i] = !consts[i].empty() && (consts[i][0] & 1);  // value mod 2
    s.const_words[n_consts] = 0;  // synthetic 0 for mul_constant by an even constant
    s.const_words[n_consts + 1] = 1;  // synthetic 1: a constant >= 2 in front of zero tests (track_unreduced_values)
    for (DevOp& d : s.ops)
      if (d.kind == TK_CONST && d.a == kSyntheticOne) d.a = n_consts + 1;
      else if (d.kind == TK_CONST && d.a == kSyntheticZero) d.a = n_consts;
    // arithmetic mod 2 on {0,1}: (a+b)%2 = xor, (a*b)%2 = and
    for (DevOp& d : s.ops) {
      if (d.kind == TK_ADD) d.kind = TK_XOR;
      else if (d.kind == TK_MUL) d.kind = TK_AND;
      else if (d.kind == TK_ADDC) d.kind = s.const_words[d.b] ? TK_NOT : TK_COPY;
      else if (d.kind == TK_MULC) {
        if (s.const_words[d.b]) d.kind = TK_COPY;
        else { d.kind = TK_CONST; d.a = n_consts; }
      }
    }
  } else {
    s.words_per_const = m.field.nwords;
    s.const_words.assign((size_t)n_consts * m.field.nwords, 0);
    for (uint32_t i = 0; i < n_consts; ++i) {
      uint32_t r[kFieldWords], mont[kFieldWords];
      m.field.reduce(consts[i], r);
      m.field.to_mont(r, mont);
      memcpy(&s.const_words[(size_t)i * m.field.nwords], mont, 4 * m.field.nwords);
    }
    // the constants whose unreduced bits are read, as the integers they are (Schedule::raw_const_of)
    for (uint32_t c : s.raw_const_of) {
      const Value& v = consts[c];
      size_t n = v.size();
      while (n > 0 && v[n - 1] == 0) --n;
      if (n > 4 * (size_t)m.field.nwords)
        throw Error("GPU backend: a constant wider than the field's limbs reaches and / xor or Evaluator::get without passing "
                    "through an arithmetic gate; the reference works on the unreduced integer there (evaluator.rs:750-752,924-933) "
                    "and this path cannot hold it");
      std::vector<uint32_t> wds(m.field.nwords, 0);
      for (size_t b = 0; b < n; ++b) wds[b / 4] |= (uint32_t)v[b] << (8 * (b % 4));
      s.const_words.insert(s.const_words.end(), wds.begin(), wds.end());
    }
  }
  return std::move(s);
}

namespace {

Schedule schedule_windows(const Tape& tape, const FieldHost& field, const ScheduleOptions& opt, const std::vector<uint32_t>& cuts) {
  StreamScheduler sch(field, opt);
  size_t drop_at = 0, ladder_at = 0;
  uint32_t lo = 0;
  const uint32_t n = (uint32_t)tape.size();
  for (size_t c = 0; c <= cuts.size(); ++c) {
    const bool final = c == cuts.size();
    const uint32_t hi = final ? n : std::min(cuts[c], n);
    TapeWindow w;
    w.lo = lo;
    w.hi = hi;
    w.kind = tape.kind.data() + lo;
    w.a = tape.a.data() + lo;
    w.b = tape.b.data() + lo;
    size_t d1 = drop_at;
    while (d1 < tape.drop_pos.size() && tape.drop_pos[d1] <= hi) ++d1;
    w.drops = tape.drop_handle.data() + drop_at;
    w.n_drops = d1 - drop_at;
    drop_at = d1;
    size_t l1 = ladder_at;
    while (l1 < tape.ladders.size() && tape.ladders[l1].result < hi) ++l1;
    w.ladders = tape.ladders.data() + ladder_at;
    w.n_ladders = l1 - ladder_at;
    ladder_at = l1;
    w.final = final;
    w.pinned = &opt.pinned;
    w.consts = &tape.consts;
    sch.add_window(w);
    lo = hi;
  }
  return sch.finish(tape.consts);
}

}  // namespace

Schedule build_schedule(const Tape& tape, const FieldHost& field, const ScheduleOptions& opt) {
  return schedule_windows(tape, field, opt, {});
}

Schedule build_schedule_windowed(const Tape& tape, const FieldHost& field, const ScheduleOptions& opt) {
  return schedule_windows(tape, field, opt, tape.cuts);
}

}  // namespace zki
