// extern "C" surface declared in include/zkgpu.h.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/zkgpu.h"
#include "affinity.hpp"
#include "engine.hpp"
#include "lds_program.hpp"
#include "evaluator.hpp"
#include "r1cs.hpp"
#include "schedule.hpp"
#include "stats.hpp"
#include "tape.hpp"
#include "validator.hpp"

using namespace zki;

// One tape window handed from the recording thread to the scheduling thread: its own copy of the ops (the tape's
// vectors keep growing -- and moving -- while the window is scheduled).
struct WindowJob {
  uint32_t lo = 0, hi = 0;
  std::vector<uint8_t> kind;
  std::vector<uint32_t> a, b, drops, pinned;
  std::vector<Tape::Ladder> ladders;
  std::vector<uint8_t> const_parity;   // GF(2): low bit of every constant recorded so far (decides what add / mul by a constant become)
  bool final = false;
};

// Streaming ingest (option "stream"): the recording thread cuts the tape into windows (tape.hpp `cuts`); a worker
// schedules each window as soon as it is complete and sends its program entries to the GPU, while the caller is still
// parsing the following messages (the reference consumes a relation message by message too, evaluator.rs:286-301).
struct StreamState {
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<WindowJob> jobs;
  bool quit = false, idle = true;
  std::string error;
  std::unique_ptr<StreamScheduler> sched;
  uint32_t next_lo = 0;
  size_t drop_at = 0, ladder_at = 0;
  uint32_t windows_done = 0;
  double busy_s = 0;     // time the worker spent scheduling + uploading
  bool upload = true;    // send finished windows to the GPU (switched off when no GPU can be opened)
  int device = -2;       // the engine's device: the first listed one, or the RECORDING thread's current device (-2: not asked yet)
};

// A finished FIELD SEGMENT of the session.  The reference takes the modulus afresh from every message header
// (evaluator.rs:232-237, :262-268): a Relation message may continue under another field characteristic, and the wires
// alive in the scope simply live on as the integers they are.  Here every field has a backend, a schedule and an engine of
// its own (kernels, wire-table stride and Montgomery constants are per field): when the modulus changes, the part recorded
// so far is put aside as a segment, and the wires of the scope are re-bound to TK_CARRY sources of the new backend, whose
// values the old engine writes out as canonical integers after its replay (Engine::carry_out).  The engines of a session
// run one after the other on one stream and share the verdict words of the first.
struct FieldSegment {
  TapeBackend backend;
  Schedule sched;
  std::unique_ptr<Engine> engine;
  std::vector<std::unique_ptr<Engine>> peers;   // option "devices": this segment's engines on the other devices (like zkgpu_session::peers)
  std::vector<uint32_t> carried_out;   // its handles whose values the next segment reads, in carry order
  std::vector<uint32_t> value_op_index;
};

struct zkgpu_session {
  TapeBackend backend;                                  // the CURRENT (last) field segment: backend, sched, engine
  std::vector<std::unique_ptr<FieldSegment>> prev;      // the segments before it, in order
  int inspect_segment = -1;                             // option "inspect_segment": which one the introspection calls describe
  Evaluator<TapeBackend> ev;
  std::unique_ptr<Engine> engine;
  // option "devices": the lanes of a batch are split over several engines, one per listed device (engine = the first,
  // peers = the others), each driven from a host thread of its own; lane_first[k] = first lane of engine k's share
  std::vector<int> devices;
  std::vector<std::unique_ptr<Engine>> peers;
  std::vector<uint32_t> lane_first;
  std::unique_ptr<CountReducer> reducer;   // RCCL, when the devices are distinct
  bool force_rccl = false;                 // option "force_rccl": zkgpu_counts always goes through RCCL, failures are errors
  uint64_t rccl_reductions = 0;            // zkgpu_counts calls answered by an RCCL all-reduce
  std::string rccl_note;                   // why the host sum was used instead (RCCL could not be loaded / initialised)
  uint32_t batch = 0;
  Schedule sched;
  bool finalized = false;
  bool engine_loaded = false;   // the engine holds the finished program (a streamed ingest opens the engine earlier)
  bool used_evaluator = false;  // messages went through the bundled Evaluator (zkgpu_ingest_* / declare_inputs)
  // Caller-driven backend (zkgpu_backend_*): the handles the caller holds are SESSION-wide numbers.  In the first field
  // segment they are tape indices; a field change (zkgpu_backend_set_field with another modulus) moves the handle space up
  // by the entries recorded so far -- handle = handle_base + tape index of the current segment -- and the wires that were
  // alive at the change (every handle not reported dropped) are re-bound: old handle -> tape index in `rebound`.
  uint32_t handle_base = 0;
  std::unordered_map<uint32_t, uint32_t> rebound;
  bool retain_all = false;
  uint32_t declared_inst = 0, declared_wit = 0;
  uint32_t lane_group = 0;
  int bool_path = 0;      // 0 auto, 1 HBM-table kernel, 2 LDS-resident kernel
  int sort_by_operand = 3;
  bool fuse = true;
  bool propagate_copies = true;
  bool pair = true;
  bool fermat = true;
  uint32_t n_streams = 2;
  bool xcd_map = true;
  int graph_mode = 0;
  uint32_t level_ops_per_wave = 1;
  uint32_t hot_waves = 0;
  uint32_t stream_window = 0;        // option "stream": tape entries per window, 0 = schedule everything at finalize
  bool stream_explicit = false;      // the caller set "stream" (otherwise GF(2) relations stream: stream_by_default)
  uint32_t sched_threads = 0;
  bool bank_aware = true;
  bool strand_lds = true, strand_prefetch = true, strand_merge = true, strand_reassociate = true, strand_split_inputs = true;
  uint32_t bool_narrow_width = 0;   // 0 = the scheduler's default
  uint32_t strand_width = 0;   // 0 = the scheduler's default
  std::unique_ptr<StreamState> stream;
  double stream_busy_s = 0;
  double parse_s = 0, record_s = 0;   // seconds spent decoding messages / recording Relation messages (ZKI_SCHED_PROFILE prints them)
  uint32_t stream_windows = 0;
  size_t n_pinned = 0;
  R1cs r1cs;                         // constraint system derived from the tape or loaded as CSR
  bool r1cs_ready = false, r1cs_on_device = false, r1cs_loaded_csr = false;
  bool r1cs_coef_classes = true;     // option "r1cs_coef_classes": combinations of coefficients 1 / -1 / small integers take the cheap row paths
  std::vector<R1csRowDev> r1cs_rows_dev;
  std::vector<R1csTermDev> r1cs_terms_dev;
  std::vector<uint32_t> r1cs_coef_words;
  uint32_t r1cs_extra_vars = 0;
  // the other two consumers of `valid-eval-metrics` (cli.rs:333-363), fed the same messages when enabled
  std::unique_ptr<Validator> validator;
  std::unique_ptr<Stats> stats;
  std::string last_error;
  std::vector<uint32_t> first_fail, flags;
  std::vector<uint32_t> value_op_index;  // k-th value-returning call -> tape index
  bool results_fresh = false;
  ~zkgpu_session();
};

zkgpu_session::~zkgpu_session() {
  if (stream) {
    {
      std::lock_guard<std::mutex> g(stream->mu);
      stream->quit = true;
    }
    stream->cv.notify_all();
    if (stream->worker.joinable()) stream->worker.join();
  }
}

namespace {

template <class F>
int guarded(zkgpu_session* s, F&& f) {
  if (!s) return -1;
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    s->last_error = e.what();
    return 1;
  } catch (...) {
    s->last_error = "unknown error";
    return 2;
  }
}

uint32_t lane_inputs(const zkgpu_session* s, bool instance) {
  const Tape& t = s->backend.tape();
  uint32_t n = instance ? std::max<uint32_t>(t.n_instance, std::max<uint32_t>(s->declared_inst, (uint32_t)s->backend.lane0_instances().size()))
                        : std::max<uint32_t>(t.n_witness, std::max<uint32_t>(s->declared_wit, (uint32_t)s->backend.lane0_witnesses().size()));
  for (const auto& seg : s->prev) n = std::max(n, instance ? seg->backend.tape().n_instance : seg->backend.tape().n_witness);
  return n;
}

// ---- field segments ----------------------------------------------------------------------------------
size_t n_segments(const zkgpu_session* s) { return s->prev.size() + 1; }
const TapeBackend& seg_backend(const zkgpu_session* s, size_t k) { return k < s->prev.size() ? s->prev[k]->backend : s->backend; }
const Schedule& seg_sched(const zkgpu_session* s, size_t k) { return k < s->prev.size() ? s->prev[k]->sched : s->sched; }
Engine* seg_engine(const zkgpu_session* s, size_t k) { return k < s->prev.size() ? s->prev[k]->engine.get() : s->engine.get(); }
// the segment the introspection calls (tape / schedule dumps, constants, modulus) describe: the last one unless told otherwise
size_t inspected(const zkgpu_session* s) {
  return (s->inspect_segment >= 0 && (size_t)s->inspect_segment < n_segments(s)) ? (size_t)s->inspect_segment : s->prev.size();
}
// bytes per input value in the caller's buffers: the limbs of the widest field of the session
uint32_t session_elem_bytes(const zkgpu_session* s) {
  if (!s->backend.field_set()) return 0;
  uint32_t w = s->backend.field().is_two ? 1 : 4 * s->backend.field().nwords;
  for (const auto& seg : s->prev) w = std::max<uint32_t>(w, 4 * seg->backend.field().nwords);
  return w;
}
// first assert sequence number of segment k (the asserts of a session are numbered through)
uint32_t seg_assert_base(const zkgpu_session* s, size_t k) { return seg_backend(s, k).assert_base(); }

// k-th value-returning backend call -> tape index (flattened wire k of IRFlattener's numbering)
void need_value_index(zkgpu_session* s) {
  auto build = [](const Tape& t, std::vector<uint32_t>& idx) {
    if (!idx.empty() || !t.size()) return;
    idx.reserve(t.n_value_ops);
    for (size_t i = t.n_rebound; i < t.size(); ++i)   // (the wires re-bound at a field change are no backend calls)
      if (t.kind[i] != TK_ASSERT && t.kind[i] != TK_CARRY) idx.push_back((uint32_t)i);
  };
  build(s->backend.tape(), s->value_op_index);
  for (auto& seg : s->prev) build(seg->backend.tape(), seg->value_op_index);
}

ScheduleOptions schedule_options(const zkgpu_session* s, bool retain_all) {
  ScheduleOptions opt;
  opt.retain_all = retain_all;
  opt.sort_by_operand = s->sort_by_operand;
  opt.pair = s->pair;
  opt.fermat = s->fermat;
  opt.fuse = s->fuse;
  opt.propagate_copies = s->propagate_copies;
  opt.threads = s->sched_threads;
  opt.bank_aware = s->bank_aware;
  opt.strand_lds = s->strand_lds;
  opt.strand_prefetch = s->strand_prefetch;
  opt.strand_merge = s->strand_merge;
  opt.strand_reassociate = s->strand_reassociate;
  opt.strand_split_inputs = s->strand_split_inputs;
  if (s->bool_narrow_width) opt.bool_narrow_width = s->bool_narrow_width;
  if (s->strand_width) opt.strand_width = s->strand_width;
  return opt;
}

void configure_engine(zkgpu_session* s, Engine* e) {
  e->set_bool_path(s->bool_path);
  e->set_lane_group(s->lane_group);
  e->set_streams(s->n_streams);
  e->set_xcd_map(s->xcd_map);
  e->set_graph_mode(s->graph_mode);
  e->set_level_ops_per_wave(s->level_ops_per_wave);
  e->set_hot_waves(s->hot_waves);
}

// ---- streaming ingest ------------------------------------------------------------------------------
void stream_worker(zkgpu_session* s) {
  StreamState& st = *s->stream;
  for (;;) {
    WindowJob job;
    {
      std::unique_lock<std::mutex> lk(st.mu);
      st.idle = st.jobs.empty();
      if (st.idle) st.cv.notify_all();
      st.cv.wait(lk, [&] { return st.quit || !st.jobs.empty(); });
      if (st.jobs.empty()) return;  // quit
      job = std::move(st.jobs.front());
      st.jobs.pop_front();
      st.idle = false;
    }
    if (!st.error.empty()) continue;  // an earlier window failed: drain
    const auto t0 = std::chrono::steady_clock::now();
    try {
      TapeWindow w;
      w.lo = job.lo;
      w.hi = job.hi;
      w.kind = job.kind.data();
      w.a = job.a.data();
      w.b = job.b.data();
      w.drops = job.drops.data();
      w.n_drops = job.drops.size();
      w.ladders = job.ladders.data();
      w.n_ladders = job.ladders.size();
      w.final = job.final;
      w.pinned = &job.pinned;
      if (!job.const_parity.empty()) w.const_parity = &job.const_parity;
      const WindowResult r = st.sched->add_window(w);
      if (st.upload) {
        // the window's entries go to HBM now; without a GPU (the CPU test tier) they are sent by the first replay call
        try {
          if (!s->engine) s->engine.reset(new Engine(st.device));
        } catch (const std::exception&) {
          st.upload = false;
        }
        if (st.upload) {
          const Schedule& p = st.sched->partial();
          if (p.fused) s->engine->upload_window(p.ops2.data() + r.first_op, r.n_ops, sizeof(DevOp2));
          else s->engine->upload_window(p.ops.data() + r.first_op, r.n_ops, sizeof(DevOp));
        }
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> g(st.mu);
      st.error = e.what();
    }
    st.busy_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ++st.windows_done;
  }
}

// the tape ops [next_lo, hi) as a job of their own (runs on the recording thread)
void stream_enqueue(zkgpu_session* s, uint32_t hi, bool final) {
  StreamState& st = *s->stream;
  const Tape& t = s->backend.tape();
  // the worker thread opens the engine: on the device the caller works on, not on the worker's own default
  if (st.device == -2) st.device = s->devices.empty() ? current_device() : s->devices[0];
  WindowJob job;
  job.lo = st.next_lo;
  job.hi = hi;
  job.final = final;
  job.kind.assign(t.kind.begin() + job.lo, t.kind.begin() + hi);
  job.a.assign(t.a.begin() + job.lo, t.a.begin() + hi);
  job.b.assign(t.b.begin() + job.lo, t.b.begin() + hi);
  size_t d1 = st.drop_at;
  while (d1 < t.drop_pos.size() && t.drop_pos[d1] <= hi) ++d1;
  job.drops.assign(t.drop_handle.begin() + st.drop_at, t.drop_handle.begin() + d1);
  st.drop_at = d1;
  size_t l1 = st.ladder_at;
  while (l1 < t.ladders.size() && t.ladders[l1].result < hi) ++l1;
  job.ladders.assign(t.ladders.begin() + st.ladder_at, t.ladders.begin() + l1);
  st.ladder_at = l1;
  if (final) s->ev.values().for_each([&](WireId, const TapeWire& w) { job.pinned.push_back(w.h); });
  if (s->backend.field().is_two) {
    job.const_parity.resize(t.consts.size() + 1, 0);   // (+ 1: never empty, so the window knows the table is there)
    for (size_t c = 0; c < t.consts.size(); ++c) job.const_parity[c] = !t.consts[c].empty() && (t.consts[c][0] & 1);
  }
  st.next_lo = hi;
  {
    std::lock_guard<std::mutex> g(st.mu);
    st.jobs.push_back(std::move(job));
    st.idle = false;
  }
  st.cv.notify_all();
}

// TapeBackend's cut hook: a window of the tape is complete
void stream_cut(void* arg) {
  zkgpu_session* s = (zkgpu_session*)arg;
  if (!s->stream) {
    s->stream.reset(new StreamState());
    // GF(2): the 16-byte entries the windows hold are the HBM-table kernel's; the LDS-resident kernel runs a program built
    // from the finished schedule (Engine::load_program), so nothing is sent ahead unless that kernel is asked for
    if (s->backend.field().is_two && s->bool_path != 1) s->stream->upload = false;
    s->stream->sched.reset(new StreamScheduler(s->backend.field(), schedule_options(s, false)));
    s->stream->worker = std::thread(stream_worker, s);
  }
  stream_enqueue(s, s->backend.tape().cuts.back(), false);
}

void stream_wait(zkgpu_session* s) {
  StreamState& st = *s->stream;
  std::unique_lock<std::mutex> lk(st.mu);
  st.cv.wait(lk, [&] { return st.jobs.empty() && st.idle; });
}

std::vector<Engine*> all_engines(zkgpu_session* s) {
  std::vector<Engine*> v;
  if (s->engine) v.push_back(s->engine.get());
  for (auto& p : s->peers) v.push_back(p.get());
  return v;
}

// every engine an option setter has to reach: those and the engines of the field segments before the current one
std::vector<Engine*> configurable_engines(zkgpu_session* s) {
  std::vector<Engine*> v = all_engines(s);
  for (auto& seg : s->prev) {
    if (seg->engine) v.push_back(seg->engine.get());
    for (auto& p : seg->peers) v.push_back(p.get());
  }
  return v;
}

// engines that hold lanes of the current batch
std::vector<size_t> active_engines(const zkgpu_session* s) {
  std::vector<size_t> v;
  for (size_t k = 0; k + 1 < s->lane_first.size(); ++k)
    if (s->lane_first[k + 1] > s->lane_first[k]) v.push_back(k);
  return v;
}

// f(k) for every listed engine, each on a host thread of its own when there are several (a replay is hundreds of
// kernel launches: one thread feeding eight GPUs would be the bottleneck); the first exception is re-thrown
template <class F>
void per_engine(const std::vector<size_t>& which, F&& f) {
  if (which.size() <= 1) {
    for (size_t k : which) f(k);
    return;
  }
  std::vector<std::string> failed(which.size());
  std::vector<std::thread> pool;
  for (size_t t = 0; t < which.size(); ++t)
    pool.emplace_back([&, t] {
      try {
        f(which[t]);
      } catch (const std::exception& e) {
        failed[t] = e.what();
        if (failed[t].empty()) failed[t] = "error";
      }
    });
  for (auto& th : pool) th.join();
  for (const std::string& m : failed)
    if (!m.empty()) throw std::runtime_error(m);
}

// The engine (and with it the HIP runtime / a GPU) is only touched by the
// replay entry points; recording and scheduling are host work.
void need_engine(zkgpu_session* s) {
  if (!s->finalized) throw std::runtime_error("zkgpu_finalize() has not been called");
  if (!s->engine || !s->engine_loaded) {
    const int dev0 = s->devices.empty() ? current_device() : s->devices[0];   // -1 without a GPU: Engine() says so
    // a streamed ingest opened the first engine already (on the current device)
    std::unique_ptr<Engine> e = (s->engine && s->engine->device() == dev0) ? std::move(s->engine) : std::unique_ptr<Engine>(new Engine(dev0));
    s->engine.reset();
    s->peers.clear();
    s->reducer.reset();
    for (auto& seg : s->prev) {
      seg->engine.reset();
      seg->peers.clear();
    }
    const uint32_t n_inst = lane_inputs(s, true), n_wit = lane_inputs(s, false);
    auto setup = [&](Engine* x) {
      configure_engine(s, x);
      x->set_writeback(s->n_pinned != 0);
      x->load_program(s->sched, s->backend.field(), n_inst, n_wit);
      if (s->r1cs_extra_vars) x->reserve_extra_slots(s->r1cs_extra_vars);
    };
    if (!s->prev.empty()) {
      // several field segments: per device one engine per field, chained on the stream of the first; the caller's input
      // buffers have the width of the widest field.  With option "devices" every device gets a chain of its own for its
      // share of the lanes (the programs are replicated, as for a single field).
      const uint32_t stride = session_elem_bytes(s);
      const size_t n = n_segments(s);
      const size_t n_dev = std::max<size_t>(1, s->devices.size());
      for (auto& seg : s->prev) seg->peers.clear();
      for (size_t d = 0; d < n_dev; ++d) {
        const int dev = s->devices.empty() ? dev0 : s->devices[d];
        std::vector<Engine*> chain;
        for (size_t k = 0; k < n; ++k) {
          std::unique_ptr<Engine> x(new Engine(dev));   // (never an engine of an earlier chain: it would still run on that chain's stream)
          configure_engine(s, x.get());
          x->set_writeback(k == n - 1 && s->n_pinned != 0);
          x->set_input_stride(stride);
          const Tape& t = seg_backend(s, k).tape();
          x->load_program(seg_sched(s, k), seg_backend(s, k).field(), n_inst, n_wit, t.n_carry, k ? seg_backend(s, k - 1).field().nwords : 0);
          chain.push_back(x.get());
          if (k < n - 1) {
            if (d == 0) s->prev[k]->engine = std::move(x);
            else s->prev[k]->peers.push_back(std::move(x));
          } else {
            if (d == 0) s->engine = std::move(x);
            else s->peers.push_back(std::move(x));
          }
        }
        for (size_t k = 0; k < n; ++k) chain[k]->chain_to(chain[0], k == 0, k == n - 1);
      }
      s->lane_first.assign(s->peers.size() + 2, 0);
      s->engine_loaded = true;
      return;
    }
    setup(e.get());
    s->engine = std::move(e);
    for (size_t k = 1; k < s->devices.size(); ++k) {
      s->peers.emplace_back(new Engine(s->devices[k]));
      setup(s->peers.back().get());
    }
    s->lane_first.assign(s->peers.size() + 2, 0);
    s->engine_loaded = true;
  }
}

// (several field segments) the chain of device `d` of the session (0: the first listed / the current device)
std::vector<Engine*> chain_engines(zkgpu_session* s, size_t d = 0) {
  std::vector<Engine*> v;
  for (auto& seg : s->prev) v.push_back(d == 0 ? seg->engine.get() : seg->peers.at(d - 1).get());
  v.push_back(d == 0 ? s->engine.get() : s->peers.at(d - 1).get());
  return v;
}
// engine of segment g on device d
Engine* seg_engine_on(zkgpu_session* s, size_t g, size_t d) {
  if (g < s->prev.size()) return d == 0 ? s->prev[g]->engine.get() : s->prev[g]->peers.at(d - 1).get();
  return d == 0 ? s->engine.get() : s->peers.at(d - 1).get();
}

void single_segment_only(const zkgpu_session* s, const char* what) {
  if (!s->prev.empty())
    throw std::runtime_error(std::string(what) + " is not available for a relation whose field characteristic changes between messages");
}

void single_device_only(const zkgpu_session* s, const char* what) {
  if (!s->peers.empty()) throw std::runtime_error(std::string(what) + " is not available with several devices (option \"devices\")");
}

// lanes of a batch over the engines: contiguous shares of whole lane blocks
void split_lanes(zkgpu_session* s, uint32_t batch) {
  const size_t n = s->peers.size() + 1;
  const uint32_t unit = s->backend.field().is_two ? 4096u : 64u;
  const uint32_t blocks = (batch + unit - 1) / unit;
  s->lane_first.assign(n + 1, 0);
  for (size_t k = 0; k <= n; ++k) s->lane_first[k] = std::min<uint64_t>(batch, (uint64_t)unit * ((uint64_t)blocks * k / n));
  s->batch = batch;
}

// {satisfied, failed} of the whole batch: RCCL all-reduce over the engines' device counters when they sit on distinct
// devices, the sum of the per-engine counters otherwise (several engines on one device: the rehearsal of the lane split)
void total_counts(zkgpu_session* s, uint64_t out[2]) {
  const std::vector<size_t> act = active_engines(s);
  std::vector<Engine*> eng = all_engines(s);
  auto host_sum = [&] {
    out[0] = out[1] = 0;
    for (size_t k : act.empty() ? std::vector<size_t>{0} : act) {
      uint64_t c[2] = {0, 0};
      eng[k]->download(nullptr, nullptr, c);
      out[0] += c[0];
      out[1] += c[1];
    }
  };
  bool distinct = act.size() == eng.size();
  for (size_t i = 0; i < eng.size() && distinct; ++i)
    for (size_t j = 0; j < i; ++j)
      if (eng[i]->device() == eng[j]->device()) distinct = false;
  // RCCL: several engines on distinct devices, or -- option "force_rccl" -- also the one-engine session (a communicator of
  // one rank: the same ncclCommInitAll + ncclAllReduce calls, on a box with one GPU)
  const bool want_rccl = distinct && (act.size() > 1 || s->force_rccl);
  if (s->force_rccl && !distinct)
    throw std::runtime_error("force_rccl: RCCL needs every engine on a device of its own and lanes on each of them");
  if (!want_rccl) {
    host_sum();
    return;
  }
  try {
    if (!s->reducer) s->reducer.reset(new CountReducer(eng));
    s->reducer->all_reduce(out);
    ++s->rccl_reductions;
  } catch (const std::exception& e) {
    if (s->force_rccl) throw;
    // the counts themselves are exact either way: without a usable RCCL they are summed on the host, and the reason is kept
    s->reducer.reset();
    s->rccl_note = e.what();
    host_sum();
  }
}

void stream_cut(void* arg);

// The Relation message about to be ingested names another field characteristic than the backend works in: close the
// current field segment and open the next (FieldSegment above).  Runs between two messages, so the top-level scope is
// the only one that exists.
// `live`: the tape handles (of the current backend) of the wires that live on, in the caller's order; returns the handles
// they have in the new backend, in the same order.
std::vector<uint32_t> switch_field_core(zkgpu_session* s, const Value& modulus, uint32_t degree, bool is_boolean, const std::vector<uint32_t>& live) {
  FieldHost next;
  next.init(modulus);
  // Between GF(2) and another field the reference carries the wires over as the integers they are (evaluator.rs:232-237);
  // bit-packed wires cannot be: such a session keeps its GF(2) segments as integers too (the any-modulus kernels) --
  // the one that has been recorded (its tape does not depend on the representation) as well as a new one.
  const bool from_gf2 = s->backend.field().is_two, to_gf2 = next.is_two;
  if (s->r1cs_ready) throw std::runtime_error("GPU backend: a field change between Relation messages is not available for R1CS sessions");
  if (s->stream) {   // a streamed schedule of the old segment: dropped, the segment is scheduled at finalize like the others
    stream_wait(s);
    {
      std::lock_guard<std::mutex> g(s->stream->mu);
      s->stream->quit = true;
    }
    s->stream->cv.notify_all();
    s->stream->worker.join();
    s->stream.reset();
    s->engine.reset();
  }
  s->backend.set_window(0, nullptr, nullptr);
  if (from_gf2) s->backend.use_generic_field();
  // The wires of the scope live on as the integers they hold (evaluator.rs:232-237).  What a wire holds is known from the
  // old tape: behind copies alone it is still the instance / witness value or the constant it started as -- possibly
  // >= the old characteristic, since PlaintextBackend never reduces those (evaluator.rs:862-864,896-898,940-946) -- and
  // the new segment reads that input or constant itself, under its own field; anything a gate has produced is a
  // canonical value of the old field and travels through the carry stream.
  std::unique_ptr<FieldSegment> seg(new FieldSegment());
  struct Origin {
    uint8_t kind;      // TK_INSTANCE / TK_WITNESS: position `at`; TK_CONST: constant `at` of the old tape; TK_CARRY: carried
    uint32_t at;
  };
  std::vector<Origin> origin;
  {
    const Tape& old = s->backend.tape();
    std::unordered_map<uint32_t, uint32_t> carried_index;   // (two live handles of one value share a carry position)
    for (uint32_t h0 : live) {
      if (h0 >= old.size() || old.kind[h0] == TK_ASSERT) throw std::runtime_error("GPU backend: a wire that lives on across the field change is not a value of the session");
      uint32_t h = h0;
      while (old.kind[h] == TK_COPY) h = old.a[h];
      if (old.kind[h] == TK_INSTANCE || old.kind[h] == TK_WITNESS || old.kind[h] == TK_CONST) {
        origin.push_back(Origin{old.kind[h], old.a[h]});
      } else {
        auto it = carried_index.find(h0);
        if (it == carried_index.end()) {
          it = carried_index.emplace(h0, (uint32_t)seg->carried_out.size()).first;
          seg->carried_out.push_back(h0);
        }
        origin.push_back(Origin{TK_CARRY, it->second});
      }
    }
  }
  const uint32_t assert_base = s->backend.assert_base() + (uint32_t)s->backend.tape().assert_op.size();
  TapeBackend fresh;
  fresh.adopt_streams(s->backend);
  seg->backend = std::move(s->backend);
  s->backend = std::move(fresh);
  if (to_gf2) s->backend.use_generic_field();
  s->backend.set_assert_base(assert_base);
  s->backend.set_field(modulus, degree, is_boolean);
  // ... and re-bind them, in the same order
  std::vector<uint32_t> out;
  out.reserve(live.size());
  std::unordered_map<uint32_t, uint32_t> carry_handle;
  for (const Origin& o : origin) {
    uint32_t h;
    if (o.kind == TK_CARRY) {
      auto it = carry_handle.find(o.at);
      if (it == carry_handle.end()) it = carry_handle.emplace(o.at, s->backend.h_carry(o.at)).first;
      h = it->second;
    } else if (o.kind == TK_CONST) {
      h = s->backend.h_constant(TapeBackend::literal_bytes(seg->backend.tape().consts[o.at]));
    } else {
      h = s->backend.h_input_at(o.kind, o.at);
    }
    out.push_back(h);
  }
  s->backend.end_rebinding();
  s->value_op_index.clear();
  s->prev.push_back(std::move(seg));
  return out;
}

void switch_field(zkgpu_session* s, const Header& header, bool is_boolean) {
  if (header.field_degree != 1) return;   // (set_field reports it, with the reference's text)
  // detach the wires of the (top-level, the only one between messages) scope from the old backend -- no drop record: the
  // old segment keeps them readable -- and bind them to what they are in the new one
  std::vector<WireId> ids;
  std::vector<uint32_t> live;
  s->ev.values().for_each([&](WireId id, const TapeWire& w) {
    ids.push_back(id);
    live.push_back(w.h);
  });
  const std::vector<uint32_t> fresh = switch_field_core(s, header.field_characteristic, header.field_degree, is_boolean, live);
  size_t k = 0;
  s->ev.values_mut().for_each_mut([&](WireId id, TapeWire& w) {
    if (k >= ids.size() || ids[k] != id) throw std::runtime_error("GPU backend: the scope changed while a field segment was opened");
    w.h = kNoWire;      // (no drop record in the old backend)
    w.owner = nullptr;
    w = TapeWire(fresh[k], &s->backend);
    ++k;
  });
}

// ---- caller-driven backend: session-wide handles ------------------------------------------------------------------
uint32_t local_handle(const zkgpu_session* s, uint32_t h) {
  if (h >= s->handle_base) return h - s->handle_base;
  auto it = s->rebound.find(h);
  if (it == s->rebound.end())
    throw std::runtime_error("GPU backend: wire " + std::to_string(h) + " was dropped before the field characteristic changed (or never existed)");
  return it->second;
}
uint32_t global_handle(const zkgpu_session* s, uint32_t local) { return s->handle_base + local; }

// zkgpu_backend_set_field with another modulus than the backend works in (evaluator.rs:262-268: the reference's Evaluator
// calls set_field for every Relation message, with whatever modulus the header holds): a new field segment.  The wires
// that live on are the caller's handles that have not been dropped (zkgpu_backend_drop = `impl Drop` of its Wire type).
void switch_field_caller_driven(zkgpu_session* s, const Value& modulus, uint32_t degree, bool is_boolean) {
  if (degree != 1) {
    s->backend.set_field(modulus, degree, is_boolean);   // (reports it, with the reference's text)
    return;
  }
  const Tape& t = s->backend.tape();
  std::vector<uint8_t> dropped(t.size(), 0);
  for (uint32_t h : t.drop_handle)
    if (h < t.size()) dropped[h] = 1;
  // session-wide handle of every live wire, and its tape index: the values of this segment ...
  std::vector<uint32_t> globals, live;
  for (const auto& kv : s->rebound)
    if (!dropped[kv.second]) {
      globals.push_back(kv.first);
      live.push_back(kv.second);
    }
  for (uint32_t i = t.n_rebound; i < t.size(); ++i)
    if (t.kind[i] != TK_ASSERT && !dropped[i]) {
      globals.push_back(s->handle_base + i);
      live.push_back(i);
    }
  if (live.size() > (1u << 20))
    throw std::runtime_error("GPU backend: " + std::to_string(live.size()) + " wires are alive at the field change: a caller-driven backend must "
                             "report the wires it lets go of (zkgpu_backend_drop, `impl Drop` of its Wire type) for its field to change");
  const uint32_t next_base = s->handle_base + (uint32_t)t.size();
  const std::vector<uint32_t> fresh = switch_field_core(s, modulus, degree, is_boolean, live);
  s->rebound.clear();
  for (size_t k = 0; k < globals.size(); ++k) s->rebound[globals[k]] = fresh[k];
  // the new segment's own handles start behind every handle issued so far; its tape starts with the re-bound entries
  s->handle_base = next_base;
}
// A byte stream of several messages (a file of a workspace holds up to hundreds of <= 100k-gate Relation messages) is
// DECODED on a helper thread while the messages before are recorded: decoding a 10 M-gate relation takes about as long as
// recording it (FlatBuffers tables into gates: 0.28 s; gates into the tape: 0.33 s on the bench box).  The consumers see
// the messages in order, one at a time, exactly as without the helper; what a message that does not decode does is decided
// when its turn comes.
struct DecodedMessage {
  Message msg;
  std::exception_ptr failure;   // read_message threw
  std::string what;
  double seconds = 0;
};
class MessageDecoder {
 public:
  MessageDecoder(const uint8_t* data, const std::vector<std::pair<size_t, size_t>>& parts) : data_(data), parts_(parts) {
    consumer_cpu_.store(sched_getcpu());
    if (parts_.size() > 1) worker_ = std::thread([this] { run(); });
  }
  ~MessageDecoder() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    if (worker_.joinable()) worker_.join();
  }
  // message k (asked for in order)
  DecodedMessage next(size_t k) {
    if (!worker_.joinable()) return decode(k);
    consumer_cpu_.store(sched_getcpu(), std::memory_order_relaxed);   // (the decoder follows: affinity.hpp)
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !ready_.empty(); });
    DecodedMessage d = std::move(ready_.front());
    ready_.pop_front();
    lk.unlock();
    cv_.notify_all();
    return d;
  }

 private:
  DecodedMessage decode(size_t k) {
    DecodedMessage d;
    const auto t0 = std::chrono::steady_clock::now();
    try {
      d.msg = read_message(data_ + parts_[k].first, parts_[k].second);
    } catch (const std::exception& e) {
      d.failure = std::current_exception();
      d.what = e.what();
    }
    d.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return d;
  }
  void run() {
    FollowCpu place;
    for (size_t k = 0; k < parts_.size(); ++k) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || ready_.size() < 2; });   // (two decoded messages ahead at most: ~10 MB)
        if (stop_) return;
      }
      place.follow(consumer_cpu_.load(std::memory_order_relaxed));
      DecodedMessage d = decode(k);
      {
        std::lock_guard<std::mutex> g(mu_);
        ready_.push_back(std::move(d));
      }
      cv_.notify_all();
    }
  }
  const uint8_t* data_;
  const std::vector<std::pair<size_t, size_t>>& parts_;
  std::thread worker_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<DecodedMessage> ready_;
  std::atomic<int> consumer_cpu_{-1};
  bool stop_ = false;
};

// "stream" not set by the caller: a relation over GF(2) is streamed anyway -- its windows end at the seams between
// dependency levels and the program of the LDS-resident kernel comes out byte for byte as at finalize
// (tests/test_stream.py, tests/test_full_size.py), so there is nothing to lose and 0.4 s of the C4 relation's 1.1 s to first
// verdict to gain; for the other fields a streamed schedule replays 1-36 % slower (windows limit fusion and strands:
// profiles/r04_tuning_sweeps.txt) and stays the caller's choice.  Called in front of the first thing recorded.
void stream_by_default(zkgpu_session* s, const Value& modulus) {
  if (s->stream_explicit || s->stream_window || s->finalized || s->backend.field_set() || s->backend.tape().size()) return;
  size_t n = modulus.size();
  while (n > 0 && modulus[n - 1] == 0) --n;
  if (n != 1 || modulus[0] != 2) return;
  s->stream_window = 131072u;
  s->backend.set_window(s->stream_window, stream_cut, s);
}

void ingest_stream(zkgpu_session* s, const uint8_t* data, size_t len) {
  const bool side_consumers = s->validator || s->stats;
  const auto parts = split_messages(data, len);
  MessageDecoder decoder(data, parts);
  for (size_t k = 0; k < parts.size(); ++k) {
    const auto& m = parts[k];
    if (s->ev.has_error() && !side_consumers) return;
    // peek the message type: Instance / Witness messages become lane 0's streams
    DecodedMessage dec = decoder.next(k);
    s->parse_s += dec.seconds;
    if (dec.failure) {
      // `let msg = msg?;` ends valid-eval-metrics before any report (cli.rs:345-346)
      if (side_consumers) std::rethrow_exception(dec.failure);
      // Evaluator::from_messages unwraps (evaluator.rs:193): route through the latch
      s->ev.ingest_buffer(data + m.first, m.second, s->backend);
      continue;
    }
    Message& msg = dec.msg;
    if (s->validator) s->validator->ingest_message(msg);
    if (s->stats) s->stats->ingest_message(msg);
    if (s->ev.has_error()) continue;  // first error latches (evaluator.rs:213-222)
    if (msg.kind == Message::IsInstance) {
      s->ev.set_modulus(msg.instance.header.field_characteristic);
      for (const Value& v : msg.instance.common_inputs) s->ev.push_instance(s->backend.import_instance(v));
    } else if (msg.kind == Message::IsWitness) {
      s->ev.set_modulus(msg.witness.header.field_characteristic);
      for (const Value& v : msg.witness.short_witness) s->ev.push_witness(s->backend.import_witness(v));
    } else {
      if (msg.kind == Message::IsRelation && s->backend.field_set() && !s->ev.has_error()) {
        // a new modulus opens a new field segment (evaluator.rs:232-237, :262-268); what FieldHost cannot take
        // (an even modulus, more than 512 bits, zero) is left to set_field, which reports it with the reference's text
        // (the significant little-endian bytes of the header's characteristic against the one in use: no field constants
        // are derived just to compare; what FieldHost cannot take -- zero, one, more than 4096 bits -- is reported by
        // set_field, with the reference's text, from inside the new segment)
        const Value& hv = msg.relation.header.field_characteristic;
        const Value& cv = s->backend.modulus();
        size_t hn = hv.size(), cn = cv.size();
        while (hn > 0 && hv[hn - 1] == 0) --hn;
        while (cn > 0 && cv[cn - 1] == 0) --cn;
        bool differs = hn != cn || memcmp(hv.data(), cv.data(), hn) != 0;
        if (differs) {
          try {
            FieldHost f;
            f.init(hv);
          } catch (const std::exception&) {
            differs = false;   // left to set_field
          }
        }
        if (differs) switch_field(s, msg.relation.header, mask::contains_feature(msg.relation.gate_mask, mask::BOOL));
      }
      if (msg.kind == Message::IsRelation) stream_by_default(s, msg.relation.header.field_characteristic);
      const auto t_rec = std::chrono::steady_clock::now();
      s->ev.ingest_message(msg, s->backend);
      s->record_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_rec).count();
    }
  }
}

std::string join_lines(const std::vector<std::string>& v) {
  std::string out;
  for (size_t i = 0; i < v.size(); ++i) {
    if (i) out.push_back('\n');
    out += v[i];
  }
  return out;
}

size_t copy_out(const std::string& s, char* buf, size_t cap) {
  if (buf && cap) {
    const size_t n = std::min(cap - 1, s.size());
    memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return s.size();
}

void fetch_results(zkgpu_session* s) {
  need_engine(s);
  if (s->results_fresh) return;
  uint64_t counts[2];
  if (s->peers.empty()) {
    s->engine->download(&s->first_fail, &s->flags, counts);
  } else {
    s->first_fail.clear();
    s->flags.clear();
    std::vector<Engine*> eng = all_engines(s);
    for (size_t k : active_engines(s)) {   // shares are contiguous and in lane order
      std::vector<uint32_t> ff, fl;
      eng[k]->download(&ff, &fl, counts);
      s->first_fail.insert(s->first_fail.end(), ff.begin(), ff.end());
      s->flags.insert(s->flags.end(), fl.begin(), fl.end());
    }
  }
  s->results_fresh = true;
}

// out[lane][k][elem] of the listed slots for every lane of the batch, gathered from the engines that hold the lanes
void dump_slots_all(zkgpu_session* s, const std::vector<uint32_t>& slots, std::vector<uint8_t>* out) {
  if (s->peers.empty()) {
    s->engine->dump_slots(slots, out);
    return;
  }
  out->clear();
  std::vector<Engine*> eng = all_engines(s);
  for (size_t k : active_engines(s)) {
    std::vector<uint8_t> part;
    eng[k]->dump_slots(slots, &part);
    out->insert(out->end(), part.begin(), part.end());
  }
}

// pad one little-endian Value into a fixed-width slot.  A value that does not fit is >= p: its residue goes in
// (what every arithmetic gate would make of it) and `oversize` tells the caller, who flags the lane if the position is
// one where the residue will not do (Schedule::strict_instance / strict_witness).
void put_value(const Value& v, uint8_t* dst, uint32_t width, const FieldHost& f, bool* oversize) {
  size_t n = v.size();
  while (n > 0 && v[n - 1] == 0) --n;
  memset(dst, 0, width);
  if (n > width) {
    *oversize = true;
    if (f.is_two) {
      dst[0] = v[0] & 1;
    } else {
      uint32_t r[kFieldWords];
      f.reduce(v, r);
      memcpy(dst, r, std::min<size_t>(width, sizeof r));
    }
    return;
  }
  memcpy(dst, v.data(), n);
}

}  // namespace

// ---- R1CS ---------------------------------------------------------------------------------------
namespace {

// coefficient pool -> Montgomery words; coefficient 1 -> 0xFFFFFFFF (no multiply), 0 -> term dropped
struct CoefMap {
  std::vector<uint32_t> index;  // per host coefficient: device index, 0xFFFFFFFF (one) or 0xFFFFFFFE (zero)
  std::vector<uint32_t> small;  // ... as a signed integer, sign << 31 | magnitude, where the magnitude is below 2^31
                                // (the smaller of c and p - c); 0: not such a coefficient
  std::vector<uint32_t> words;
};
CoefMap device_coefs(const std::vector<Value>& coefs, const FieldHost& f) {
  CoefMap m;
  m.index.resize(coefs.size());
  m.small.assign(coefs.size(), 0);
  uint32_t pw[kFieldWords] = {0};
  memcpy(pw, f.p, sizeof(uint32_t) * std::min<size_t>(f.nwords, kFieldWords));
  for (size_t i = 0; i < coefs.size(); ++i) {
    uint32_t r[kFieldWords], mont[kFieldWords];
    f.reduce(coefs[i], r);
    bool zero = true, one = r[0] == 1, single = true;
    for (int k = 0; k < kFieldWords; ++k) {
      zero &= r[k] == 0;
      if (k) one &= r[k] == 0;
      if (k) single &= r[k] == 0;
    }
    if (zero) { m.index[i] = 0xFFFFFFFEu; continue; }
    // p - c in one word?
    uint32_t neg0 = 0;
    bool neg_single = true;
    {
      uint64_t b = 0;
      for (int k = 0; k < kFieldWords; ++k) {
        const uint64_t y = (uint64_t)pw[k] - r[k] - b;
        b = (y >> 63) & 1;
        if (k == 0) neg0 = (uint32_t)y;
        else neg_single &= (uint32_t)y == 0;
      }
    }
    const bool pos_ok = single && r[0] < 0x80000000u, neg_ok = neg_single && neg0 != 0 && neg0 < 0x80000000u;
    if (pos_ok && (!neg_ok || r[0] <= neg0)) m.small[i] = r[0];
    else if (neg_ok) m.small[i] = 0x80000000u | neg0;
    if (one) { m.index[i] = 0xFFFFFFFFu; continue; }
    f.to_mont(r, mont);
    m.index[i] = (uint32_t)(m.words.size() / f.nwords);
    m.words.insert(m.words.end(), mont, mont + f.nwords);
  }
  return m;
}

// rows in `var -> slot` form for the device; slot_of_var(var) returns 0xFFFFFFFF for the constant one
template <class SlotOf>
void build_device_rows(zkgpu_session* s, SlotOf&& slot_of_var) {
  const R1cs& r = s->r1cs;
  const CoefMap cm = device_coefs(r.coefs, s->backend.field());
  s->r1cs_rows_dev.clear();
  s->r1cs_terms_dev.clear();
  for (size_t row = 0; row < r.n_rows(); ++row) {
    R1csRowDev d;
    d.first = (uint32_t)s->r1cs_terms_dev.size();
    uint32_t n[3] = {0, 0, 0};
    bool b_is_one = false;
    uint32_t cls[3] = {0, 0, 0};
    for (int part = 0; part < 3; ++part) {
      const uint32_t t0 = r.row_ptr[3 * row + part], t1 = r.row_ptr[3 * row + part + 1];
      const size_t first_term = s->r1cs_terms_dev.size();
      bool all_unit = true, all_small = true, any_minus = false;
      for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t c = cm.index[r.terms[t].coef];
        if (c == 0xFFFFFFFEu) continue;  // zero coefficient
        R1csTermDev td;
        td.slot = slot_of_var(r.terms[t].var);
        td.coef = c;
        s->r1cs_terms_dev.push_back(td);
        ++n[part];
        const uint32_t sm = cm.small[r.terms[t].coef];
        all_small &= sm != 0;
        all_unit &= (sm & 0x7FFFFFFFu) == 1;
        any_minus |= sm == 0x80000001u;
      }
      if (part == 1 && n[1] == 1) {
        const R1csTermDev& last = s->r1cs_terms_dev.back();
        b_is_one = last.slot == 0xFFFFFFFFu && last.coef == 0xFFFFFFFFu;
      }
      // the coefficient class of the combination (device/args.hpp).  All coefficients 1: class full, whose terms of
      // coefficient 1 are plain additions already (and what zkgpu_r1cs_assign expects of C).
      if (s->r1cs_coef_classes && n[part] && all_small && !(all_unit && !any_minus) && !(part == 1 && b_is_one)) {
        cls[part] = all_unit ? kR1csClassUnit : kR1csClassSmall;
        size_t k = first_term;
        for (uint32_t t = t0; t < t1; ++t) {
          if (cm.index[r.terms[t].coef] == 0xFFFFFFFEu) continue;
          s->r1cs_terms_dev[k++].coef = cm.small[r.terms[t].coef];
        }
      }
    }
    if (n[0] > 255 || n[1] > 255 || n[2] > 255) throw std::runtime_error("R1CS row with more than 255 terms in one combination");
    d.counts = n[0] | (n[1] << 8) | (n[2] << 16) | ((b_is_one ? 1u : 0u) << 24) |
               (cls[0] << (24 + kR1csClassShiftA)) | (cls[1] << (24 + kR1csClassShiftB)) | (cls[2] << (24 + kR1csClassShiftC));
    s->r1cs_rows_dev.push_back(d);
  }
  s->r1cs_coef_words = cm.words;
  s->r1cs_on_device = false;
}

// zkgpu_r1cs_assign stores <a,w>*<b,w> into the slot of C's only term (r1cs_row_kernel<N, true>): every row of the
// call must have exactly that shape, and no row of the call may read a slot another row of it writes -- checked here,
// on the host, so that a row of any other shape is an error and never a stray store on the GPU.
void r1cs_check_assignable(zkgpu_session* s, uint32_t first_row, uint32_t n_rows) {
  if (!s->r1cs_ready) throw std::runtime_error("no R1CS: call zkgpu_r1cs_from_tape or zkgpu_r1cs_load_csr");
  const auto& rows = s->r1cs_rows_dev;
  const auto& terms = s->r1cs_terms_dev;
  if ((uint64_t)first_row + n_rows > rows.size()) throw std::runtime_error("zkgpu_r1cs_assign: row range out of bounds");
  std::vector<uint32_t> targets;
  targets.reserve(n_rows);
  for (uint32_t r = first_row; r < first_row + n_rows; ++r) {
    const uint32_t na = rows[r].counts & 0xFF, nb = (rows[r].counts >> 8) & 0xFF, nc = (rows[r].counts >> 16) & 0xFF;
    const std::string where = "zkgpu_r1cs_assign: row " + std::to_string(r);
    if (nc != 1) throw std::runtime_error(where + " has " + std::to_string(nc) + " terms in C (exactly one variable with coefficient 1 is assignable)");
    const R1csTermDev& c = terms[rows[r].first + na + nb];
    if (c.coef != 0xFFFFFFFFu) throw std::runtime_error(where + ": the coefficient of C's variable is not 1");
    if (c.slot == 0xFFFFFFFFu) throw std::runtime_error(where + ": C is the constant one, not a variable");
    targets.push_back(c.slot);
  }
  std::sort(targets.begin(), targets.end());
  if (std::adjacent_find(targets.begin(), targets.end()) != targets.end())
    throw std::runtime_error("zkgpu_r1cs_assign: two rows of the call assign the same variable");
  for (uint32_t r = first_row; r < first_row + n_rows; ++r) {
    const uint32_t na = rows[r].counts & 0xFF, nb = (rows[r].counts >> 8) & 0xFF;
    for (uint32_t t = rows[r].first; t < rows[r].first + na + nb; ++t)
      if (terms[t].slot != 0xFFFFFFFFu && std::binary_search(targets.begin(), targets.end(), terms[t].slot))
        throw std::runtime_error("zkgpu_r1cs_assign: row " + std::to_string(r) +
                                 " reads a variable that a row of the same call assigns (split the call by dependency level)");
  }
}

void r1cs_to_device(zkgpu_session* s) {
  need_engine(s);
  if (!s->r1cs_ready) throw std::runtime_error("no R1CS: call zkgpu_r1cs_from_tape or zkgpu_r1cs_load_csr");
  if (!s->r1cs_on_device) {
    // (option "devices": the rows are replicated like the program; every engine checks the rows for its share of the lanes)
    for (Engine* e : all_engines(s)) e->r1cs_upload(s->r1cs_rows_dev, s->r1cs_terms_dev, s->r1cs_coef_words);
    s->r1cs_on_device = true;
  }
}

// the engines that hold lanes of the batch (a single engine: that one)
template <class F>
void per_active_engine(zkgpu_session* s, F&& f) {
  std::vector<Engine*> eng = all_engines(s);
  if (s->peers.empty()) {
    f(eng[0]);
    return;
  }
  per_engine(active_engines(s), [&](size_t k) { f(eng[k]); });
}

}  // namespace

extern "C" {

zkgpu_session* zkgpu_session_new(void) { return new (std::nothrow) zkgpu_session(); }
void zkgpu_session_free(zkgpu_session* s) { delete s; }
const char* zkgpu_last_error(const zkgpu_session* s) { return s ? s->last_error.c_str() : "null session"; }
const char* zkgpu_version(void) { return "zkgpu 0.1 (gfx950)"; }

// ---- ZKBackend trait -------------------------------------------------------
int zkgpu_backend_set_field(zkgpu_session* s, const uint8_t* modulus_le, size_t len, uint32_t degree, int is_boolean) {
  return guarded(s, [&] {
    const Value modulus(modulus_le, modulus_le + len);
    if (s->backend.field_set() && !s->used_evaluator) {
      // another characteristic than the one in use: the recording continues in a new field segment
      size_t n = modulus.size(), m = s->backend.modulus().size();
      while (n > 0 && modulus[n - 1] == 0) --n;
      while (m > 0 && s->backend.modulus()[m - 1] == 0) --m;
      if (n != m || memcmp(modulus.data(), s->backend.modulus().data(), n) != 0) {
        if (s->finalized) throw std::runtime_error("session already finalized");
        switch_field_caller_driven(s, modulus, degree, is_boolean != 0);
        return;
      }
    }
    stream_by_default(s, modulus);
    s->backend.set_field(modulus, degree, is_boolean != 0);
  });
}
int zkgpu_backend_copy(zkgpu_session* s, uint32_t wire, uint32_t* out) {
  return guarded(s, [&] { *out = global_handle(s, s->backend.h_copy(local_handle(s, wire))); });
}
int zkgpu_backend_constant(zkgpu_session* s, const uint8_t* v, size_t len, uint32_t* out) {
  return guarded(s, [&] { *out = global_handle(s, s->backend.h_constant(TapeBackend::from_bytes_le(Value(v, v + len)))); });
}
int zkgpu_backend_assert_zero(zkgpu_session* s, uint32_t wire, uint64_t local_wire_id) {
  return guarded(s, [&] {
    const uint32_t w = local_handle(s, wire);
    s->backend.note_assert_wire(local_wire_id);
    s->backend.h_assert_zero(w);
  });
}
int zkgpu_backend_add(zkgpu_session* s, uint32_t a, uint32_t b, uint32_t* out) {
  return guarded(s, [&] { *out = global_handle(s, s->backend.h_add(local_handle(s, a), local_handle(s, b))); });
}
int zkgpu_backend_multiply(zkgpu_session* s, uint32_t a, uint32_t b, uint32_t* out) {
  return guarded(s, [&] { *out = global_handle(s, s->backend.h_multiply(local_handle(s, a), local_handle(s, b))); });
}
int zkgpu_backend_add_constant(zkgpu_session* s, uint32_t a, const uint8_t* c, size_t len, uint32_t* out) {
  return guarded(s, [&] { *out = global_handle(s, s->backend.h_add_constant(local_handle(s, a), TapeBackend::from_bytes_le(Value(c, c + len)))); });
}
int zkgpu_backend_mul_constant(zkgpu_session* s, uint32_t a, const uint8_t* c, size_t len, uint32_t* out) {
  return guarded(s, [&] { *out = global_handle(s, s->backend.h_mul_constant(local_handle(s, a), TapeBackend::from_bytes_le(Value(c, c + len)))); });
}
int zkgpu_backend_and(zkgpu_session* s, uint32_t a, uint32_t b, uint32_t* out) {
  return guarded(s, [&] { *out = global_handle(s, s->backend.h_and(local_handle(s, a), local_handle(s, b))); });
}
int zkgpu_backend_xor(zkgpu_session* s, uint32_t a, uint32_t b, uint32_t* out) {
  return guarded(s, [&] { *out = global_handle(s, s->backend.h_xor(local_handle(s, a), local_handle(s, b))); });
}
int zkgpu_backend_not(zkgpu_session* s, uint32_t a, uint32_t* out) {
  return guarded(s, [&] { *out = global_handle(s, s->backend.h_not(local_handle(s, a))); });
}
int zkgpu_backend_ladder(zkgpu_session* s, uint64_t first_call, uint32_t base, uint32_t result) {
  return guarded(s, [&] {
    // first_call is a value of zkgpu_tape_len() (the calls of every field segment so far): the current segment's entries
    // start at the calls of the segments before it, behind its own re-bound entries
    const Tape& t = s->backend.tape();
    const uint64_t before = zkgpu_tape_len(s) - (t.size() - t.n_rebound);
    if (first_call < before || first_call - before + t.n_rebound > t.size())
      throw std::runtime_error("zkgpu_backend_ladder: not a range of recorded calls");
    const uint32_t r = local_handle(s, result), b = local_handle(s, base);
    if (r >= t.size()) throw std::runtime_error("zkgpu_backend_ladder: not a range of recorded calls");
    s->backend.h_ladder((size_t)(first_call - before + t.n_rebound), b, r);
  });
}
int zkgpu_backend_drop(zkgpu_session* s, uint32_t wire) {
  return guarded(s, [&] {
    if (wire < s->handle_base && !s->rebound.count(wire)) return;   // (a wire of an earlier field segment that did not live on: nothing to tell)
    const uint32_t w = local_handle(s, wire);
    if (w >= s->backend.tape().size()) throw std::runtime_error("zkgpu_backend_drop: not a wire of this session");
    s->backend.drop_wire(w);
    if (wire < s->handle_base) s->rebound.erase(wire);
  });
}
int zkgpu_backend_instance(zkgpu_session* s, uint32_t position, uint32_t* out) {
  return guarded(s, [&] { *out = global_handle(s, s->backend.h_instance(TapeBackend::instance_ref(position))); });
}
int zkgpu_backend_witness(zkgpu_session* s, uint32_t position, uint32_t* out) {
  return guarded(s, [&] {
    TapeElement e = TapeBackend::witness_ref(position);
    *out = global_handle(s, s->backend.h_witness(&e));
  });
}

// ---- Evaluator / Source -----------------------------------------------------
int zkgpu_ingest_messages(zkgpu_session* s, const uint8_t* data, size_t len) {
  return guarded(s, [&] {
    if (s->finalized) throw std::runtime_error("session already finalized");
    s->used_evaluator = true;
    ingest_stream(s, data, len);
  });
}

int zkgpu_ingest_paths(zkgpu_session* s, const char* const* paths, size_t n_paths) {
  return guarded(s, [&] {
    if (s->finalized) throw std::runtime_error("session already finalized");
    s->used_evaluator = true;
    std::vector<std::string> v(paths, paths + n_paths);
    Source src = Source::from_dirs_and_files(v);
    src.print_filenames = false;
    src.for_each_buffer([&](const uint8_t* p, size_t n) { ingest_stream(s, p, n); });
  });
}

int zkgpu_declare_inputs(zkgpu_session* s, uint32_t n_instance, uint32_t n_witness) {
  return guarded(s, [&] {
    if (s->finalized) throw std::runtime_error("session already finalized");
    s->used_evaluator = true;
    for (uint32_t k = 0; k < n_instance; ++k) s->ev.push_instance(TapeBackend::instance_ref(s->declared_inst + k));
    for (uint32_t k = 0; k < n_witness; ++k) s->ev.push_witness(TapeBackend::witness_ref(s->declared_wit + k));
    s->declared_inst += n_instance;
    s->declared_wit += n_witness;
  });
}

size_t zkgpu_host_violations(zkgpu_session* s, char* buf, size_t cap) {
  if (!s) return 0;
  // a tape recorded through zkgpu_backend_* belongs to the caller's own Evaluator, which also owns the
  // "Did not receive any gate to verify." bookkeeping (evaluator.rs:199-203)
  if (!s->used_evaluator) return copy_out("", buf, cap);
  return copy_out(join_lines(s->ev.get_violations()), buf, cap);
}

// Tape introspection.  A session whose relation changed its field characteristic holds one tape per field segment: the
// counts are totals over the segments, the assert wires run through them in order, and the per-entry dumps (and the
// constant pool, the schedule, the modulus) describe ONE segment: the last, or the one option "inspect_segment" names.
uint64_t zkgpu_tape_len(const zkgpu_session* s) {
  if (!s) return 0;
  if (s->inspect_segment >= 0) return seg_backend(s, inspected(s)).tape().size();
  uint64_t n = 0;
  for (size_t k = 0; k < n_segments(s); ++k) n += seg_backend(s, k).tape().size() - seg_backend(s, k).tape().n_rebound;   // (re-bound wires are no calls)
  return n;
}
uint64_t zkgpu_tape_value_ops(const zkgpu_session* s) {
  if (!s) return 0;
  uint64_t n = 0;
  for (size_t k = 0; k < n_segments(s); ++k) n += seg_backend(s, k).tape().n_value_ops;
  return n;
}
uint64_t zkgpu_tape_asserts(const zkgpu_session* s) {
  if (!s) return 0;
  uint64_t n = 0;
  for (size_t k = 0; k < n_segments(s); ++k) n += seg_backend(s, k).tape().assert_op.size();
  return n;
}
int zkgpu_tape_dump(const zkgpu_session* s, uint8_t* kinds, uint32_t* a, uint32_t* b, uint64_t cap) {
  if (!s) return -1;
  const Tape& t = seg_backend(s, inspected(s)).tape();
  if (cap < t.size()) return 1;
  if (t.size()) {
    memcpy(kinds, t.kind.data(), t.size());
    memcpy(a, t.a.data(), t.size() * 4);
    memcpy(b, t.b.data(), t.size() * 4);
  }
  return 0;
}
int zkgpu_tape_assert_wires(const zkgpu_session* s, uint64_t* local_wire_ids, uint64_t cap) {
  if (!s) return -1;
  if (cap < zkgpu_tape_asserts(s)) return 1;
  for (size_t k = 0; k < n_segments(s); ++k) {
    const Tape& t = seg_backend(s, k).tape();
    if (!t.assert_wire.empty()) memcpy(local_wire_ids, t.assert_wire.data(), t.assert_wire.size() * 8);
    local_wire_ids += t.assert_wire.size();
  }
  return 0;
}
uint32_t zkgpu_n_constants(const zkgpu_session* s) { return s ? (uint32_t)seg_backend(s, inspected(s)).tape().consts.size() : 0; }
size_t zkgpu_constant_bytes(const zkgpu_session* s, uint32_t index, uint8_t* out, size_t cap) {
  if (!s || index >= seg_backend(s, inspected(s)).tape().consts.size()) return 0;
  const Value& v = seg_backend(s, inspected(s)).tape().consts[index];
  if (out && cap >= v.size() && !v.empty()) memcpy(out, v.data(), v.size());
  return v.size();
}
int zkgpu_n_field_segments(const zkgpu_session* s) { return s ? (int)n_segments(s) : 0; }
// {carried in, first assert sequence number, 32-bit words per value, carried out} of segment k
int zkgpu_field_segment_info(const zkgpu_session* s, uint32_t k, uint32_t out[4]) {
  if (!s || k >= n_segments(s)) return 1;
  out[0] = seg_backend(s, k).tape().n_carry;
  out[1] = seg_assert_base(s, k);
  out[2] = seg_backend(s, k).field().is_two ? 0 : seg_backend(s, k).field().nwords;
  out[3] = k < s->prev.size() ? (uint32_t)s->prev[k]->carried_out.size() : 0;
  return 0;
}
// how segment k keeps a wire on the device: 0 = one bit (GF(2)), 1 = Montgomery form, 2 = the canonical residue
int zkgpu_field_representation(const zkgpu_session* s, uint32_t k) {
  if (!s || k >= n_segments(s) || !seg_backend(s, k).field_set()) return -1;
  const FieldHost& f = seg_backend(s, k).field();
  return f.is_two ? 0 : f.generic ? 2 : 1;
}
// The arithmetic of the any-modulus kernels, run on the host (the same functions the kernels call): CPU-tier tests.
int zkgpu_generic_selftest(const uint8_t* modulus_le, size_t modulus_len, int op, const uint32_t* a, const uint32_t* b,
                           uint32_t* out, uint32_t* nwords) {
  try {
    FieldHost f;
    f.init(Value(modulus_le, modulus_le + modulus_len), true);
    if (nwords) *nwords = f.nwords;
    if (!a) return 0;   // (a query for the width)
    return Engine::generic_selftest(f, op, a, b, out);
  } catch (const std::exception&) {
    return 2;
  }
}
// wire-table slots of the values segment k hands to segment k + 1, in carry order (after zkgpu_finalize)
int zkgpu_field_segment_carried(const zkgpu_session* s, uint32_t k, uint32_t* slots, uint32_t cap) {
  if (!s || !s->finalized || k >= s->prev.size() || cap < s->prev[k]->carried_out.size()) return 1;
  for (size_t q = 0; q < s->prev[k]->carried_out.size(); ++q) slots[q] = s->prev[k]->sched.slot_of[s->prev[k]->carried_out[q]];
  return 0;
}

// ---- batch replay -------------------------------------------------------------
int zkgpu_finalize(zkgpu_session* s, int retain_all) {
  return guarded(s, [&] {
    if (!s->backend.field_set()) throw std::runtime_error("no Relation ingested: the field is not set");
    static const bool profile = getenv("ZKI_SCHED_PROFILE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count() * 1e3; };
    ScheduleOptions opt = schedule_options(s, retain_all != 0);
    s->ev.values().for_each([&](WireId, const TapeWire& w) { opt.pinned.push_back(w.h); });
    s->n_pinned = opt.pinned.size();
    const Tape& t = s->backend.tape();
    if (s->stream && !opt.retain_all && !s->finalized) {
      // the windows scheduled while the messages were coming in + the rest of the tape as the final window
      stream_enqueue(s, (uint32_t)t.size(), true);
      const double t_enq = since(t_begin);
      stream_wait(s);
      const double t_wait = since(t_begin);
      if (!s->stream->error.empty()) throw std::runtime_error("streamed scheduling failed: " + s->stream->error);
      s->sched = s->stream->sched->finish(t.consts);
      if (profile) fprintf(stderr, "[finalize] streamed: enqueue of the tail %.1f, wait for the worker %.1f, finish %.1f ms\n", t_enq, t_wait - t_enq, since(t_begin) - t_wait);
      s->stream_busy_s = s->stream->busy_s;
      s->stream_windows = s->stream->windows_done;
      {
        std::lock_guard<std::mutex> g(s->stream->mu);
        s->stream->quit = true;
      }
      s->stream->cv.notify_all();
      s->stream->worker.join();
      s->stream.reset();
    } else {
      if (s->stream) {  // retain_all (or a second finalize): the streamed program is of no use, schedule the tape again
        stream_wait(s);
        {
          std::lock_guard<std::mutex> g(s->stream->mu);
          s->stream->quit = true;
        }
        s->stream->cv.notify_all();
        s->stream->worker.join();
        s->stream.reset();
      }
      s->engine.reset();
      // "stream" set: the same windows a streamed ingest would have scheduled (the cuts are a property of the tape)
      s->sched = s->stream_window && !opt.retain_all && s->prev.empty() ? build_schedule_windowed(t, s->backend.field(), opt)
                                                                        : build_schedule(t, s->backend.field(), opt);
    }
    // the field segments before the current one: everything their tapes hold is known; the wires the next segment
    // carries on are the ones that must stay readable
    for (auto& seg : s->prev) {
      ScheduleOptions o = schedule_options(s, retain_all != 0);
      o.pinned = seg->carried_out;
      o.pinned_are_carried = true;
      seg->sched = build_schedule(seg->backend.tape(), seg->backend.field(), o);
      seg->engine.reset();
    }
    s->backend.set_window(0, nullptr, nullptr);
    s->retain_all = opt.retain_all;
    s->value_op_index.clear();  // built on first use (need_value_index): only trace dumps and the R1CS entry points read it
    s->engine_loaded = false;
    // host check of every index the kernels use
    const auto t_validate = std::chrono::steady_clock::now();
    const double schedule_ms = since(t_begin);
    for (size_t k = 0; k < n_segments(s); ++k)
      Engine::validate_program(seg_sched(s, k), lane_inputs(s, true), lane_inputs(s, false), seg_backend(s, k).tape().n_carry);
    if (profile)
      fprintf(stderr, "[finalize] schedule %.1f validate %.1f ms | ingest so far: decode %.1f record %.1f ms\n", schedule_ms, since(t_validate),
              s->parse_s * 1e3, s->record_s * 1e3);
    s->finalized = true;
    s->results_fresh = false;
  });
}

size_t zkgpu_input_modes(const zkgpu_session* s, int witness, uint8_t* out, size_t cap) {
  if (!s || !s->finalized) return 0;
  if (witness == 2) {   // the values carried into the inspected field segment
    const Schedule& sc = seg_sched(s, inspected(s));
    const size_t nc = seg_backend(s, inspected(s)).tape().n_carry;
    for (size_t k = 0; k < nc && k < cap && out; ++k) out[k] = k < sc.strict_carry.size() ? sc.strict_carry[k] : 0;
    return nc;
  }
  const size_t n = witness ? lane_inputs(s, false) : lane_inputs(s, true);
  for (size_t k = 0; k < n && k < cap && out; ++k) {
    uint8_t m = 0;   // (a position is consumed by one field segment; the others say 0)
    for (size_t g = 0; g < n_segments(s); ++g) {
      const std::vector<uint8_t>& v = witness ? seg_sched(s, g).strict_witness : seg_sched(s, g).strict_instance;
      if (k < v.size()) m = std::max(m, v[k]);
    }
    out[k] = m;
  }
  return n;
}

int zkgpu_stream_info(const zkgpu_session* s, double out[3]) {
  if (!s || !s->finalized) return 1;
  out[0] = (double)(s->sched.window_first_op.empty() ? 0 : s->sched.window_first_op.size() - 1);
  out[1] = (double)s->stream_windows;
  out[2] = s->stream_busy_s;
  return 0;
}

uint32_t zkgpu_elem_bytes(const zkgpu_session* s) {   // (several field segments: the limbs of the widest field)
  if (!s || !s->backend.field_set()) return 0;
  return session_elem_bytes(s);
}
uint32_t zkgpu_n_instance(const zkgpu_session* s) { return s ? lane_inputs(s, true) : 0; }
uint32_t zkgpu_n_witness(const zkgpu_session* s) { return s ? lane_inputs(s, false) : 0; }

int zkgpu_schedule_info(const zkgpu_session* s, uint64_t out[8]) {
  if (!s || !s->finalized) return 1;
  const Schedule& sc = seg_sched(s, inspected(s));
  memset(out, 0, 8 * sizeof(uint64_t));
  out[0] = sc.n_levels;
  out[1] = sc.launches.size();
  out[2] = sc.n_slots;
  out[3] = sc.max_level_width;
  for (const Launch& l : sc.launches) out[4] += l.sequential ? 1 : 0;
  out[5] = sc.fused ? sc.ops2.size() : sc.ops.size();
  out[6] = sc.const_words.size();
  out[7] = sc.words_per_const;
  return 0;
}

int zkgpu_lds_program(zkgpu_session* s, uint32_t block_rows, uint64_t sizes[6], uint16_t* ops8, uint16_t* rows,
                      uint32_t* blocks, uint32_t* chunks) {
  return guarded(s, [&] {
    if (!s->finalized) throw std::runtime_error("zkgpu_finalize() has not been called");
    if (!s->backend.field().is_two) throw std::runtime_error("the LDS-resident kernel runs GF(2) relations only");
    if (!lds_program_fits(s->sched)) throw std::runtime_error("the relation does not fit the LDS-resident GF(2) kernel");
    const LdsProgram P = build_lds_program(s->sched, {4, 6, 8, 9, 10, 12}, block_rows);
    if (sizes) {
      const uint64_t v[6] = {P.ops.size(), P.rows.size(), P.blocks.size() / 2, P.chunks.size() / 4, P.block_rows, P.n_slots};
      memcpy(sizes, v, sizeof v);
    }
    if (ops8 && !P.ops.empty()) memcpy(ops8, P.ops.data(), P.ops.size() * sizeof(zkgpu::LdsOp));
    if (rows && !P.rows.empty()) memcpy(rows, P.rows.data(), P.rows.size() * 2);
    if (blocks && !P.blocks.empty()) memcpy(blocks, P.blocks.data(), P.blocks.size() * 4);
    if (chunks && !P.chunks.empty()) memcpy(chunks, P.chunks.data(), P.chunks.size() * 4);
  });
}

int zkgpu_schedule_dump(const zkgpu_session* s, uint32_t* ops4, uint32_t* launches4, uint32_t* const_words,
                        uint32_t* slot_of) {
  if (!s || !s->finalized) return 1;
  const Schedule& sc = seg_sched(s, inspected(s));
  // ops8: {dst, kind(+operand-expression bits), a0, a1, b0, b1, 0, 0} per op, fused or not
  if (ops4) {
    uint32_t* o = ops4;
    if (sc.fused) {
      memcpy(o, sc.ops2.data(), sc.ops2.size() * sizeof(DevOp2));
    } else {
      for (size_t i = 0; i < sc.ops.size(); ++i) {
        const DevOp& d = sc.ops[i];
        const uint32_t w[8] = {d.dst, d.kind, d.a, 0, d.b, 0, 0, 0};
        memcpy(o + 8 * i, w, sizeof w);
      }
    }
  }
  if (launches4)
    for (size_t i = 0; i < sc.launches.size(); ++i) {
      launches4[4 * i + 0] = sc.launches[i].first;
      launches4[4 * i + 1] = sc.launches[i].count;
      launches4[4 * i + 2] = sc.launches[i].ops_per_wave;
      launches4[4 * i + 3] = sc.launches[i].sequential ? 1 : 0;
    }
  if (const_words && !sc.const_words.empty()) memcpy(const_words, sc.const_words.data(), sc.const_words.size() * 4);
  if (slot_of && !sc.slot_of.empty()) memcpy(slot_of, sc.slot_of.data(), sc.slot_of.size() * 4);
  return 0;
}

size_t zkgpu_schedule_strand_levels(const zkgpu_session* s, uint32_t launch, uint32_t* out, size_t cap) {
  if (!s || !s->finalized) return 0;
  const Schedule& sc = seg_sched(s, inspected(s));
  if (!sc.fused || launch >= sc.launches.size()) return 0;
  const Launch& L = sc.launches[launch];
  if (!L.sequential || !L.strand_levels) return 0;
  const size_t n = (size_t)L.strand_levels + 2;
  if (out && cap >= n) {
    for (uint32_t k = 0; k <= L.strand_levels; ++k) out[k] = sc.strand_level_ptr[L.level_ptr + k];
    out[L.strand_levels + 1] = L.lds_slots;
  }
  return n;
}

int zkgpu_set_inputs(zkgpu_session* s, const uint8_t* instances, const uint8_t* witnesses, uint32_t batch) {
  return guarded(s, [&] {
    need_engine(s);
    if (!s->prev.empty()) {
      // field segments: per device the first engine of the chain takes its share of the buffers, the others read the same
      // device copies (no hand-over overlap here: every segment reads the inputs)
      split_lanes(s, batch);
      const uint32_t w = session_elem_bytes(s);
      const size_t irow = (size_t)lane_inputs(s, true) * w, wrow = (size_t)lane_inputs(s, false) * w;
      per_engine(active_engines(s), [&](size_t d) {
        std::vector<Engine*> chain = chain_engines(s, d);
        for (Engine* e : chain) e->synchronize();
        for (Engine* e : chain) e->set_batch(s->lane_first[d + 1] - s->lane_first[d]);
        chain[0]->upload_inputs(instances ? instances + irow * s->lane_first[d] : nullptr, witnesses ? witnesses + wrow * s->lane_first[d] : nullptr);
        for (size_t k = 1; k < chain.size(); ++k) chain[k]->use_device_inputs(chain[0]->device_instances(), chain[0]->device_witnesses());
      });
      s->results_fresh = false;
      return;
    }
    split_lanes(s, batch);
    const uint32_t w = s->engine->elem_bytes();
    const size_t irow = (size_t)lane_inputs(s, true) * w, wrow = (size_t)lane_inputs(s, false) * w;
    std::vector<Engine*> eng = all_engines(s);
    per_engine(active_engines(s), [&](size_t k) {
      eng[k]->set_batch(s->lane_first[k + 1] - s->lane_first[k]);
      eng[k]->upload_inputs(instances ? instances + irow * s->lane_first[k] : nullptr,
                            witnesses ? witnesses + wrow * s->lane_first[k] : nullptr);
    });
    s->results_fresh = false;
  });
}

int zkgpu_set_inputs_device(zkgpu_session* s, const void* d_instances, const void* d_witnesses, uint32_t batch) {
  return guarded(s, [&] {
    need_engine(s);
    single_device_only(s, "zkgpu_set_inputs_device");
    split_lanes(s, batch);
    for (auto& seg : s->prev) {
      seg->engine->set_batch(batch);
      seg->engine->use_device_inputs(d_instances, d_witnesses);
    }
    s->engine->set_batch(batch);
    s->engine->use_device_inputs(d_instances, d_witnesses);
    s->results_fresh = false;
  });
}

int zkgpu_set_inputs_from_messages(zkgpu_session* s) {
  return guarded(s, [&] {
    need_engine(s);
    const uint32_t w = session_elem_bytes(s);
    const uint32_t ni = zkgpu_n_instance(s), nw = zkgpu_n_witness(s);
    std::vector<uint8_t> inst((size_t)ni * w, 0), wit((size_t)nw * w, 0);
    const auto& li = s->backend.lane0_instances();
    const auto& lw = s->backend.lane0_witnesses();
    if (ni > li.size() || nw > lw.size())
      throw std::runtime_error("the tape consumes more instance/witness values than the ingested messages hold");
    // A value too wide for the buffer goes in as its residue -- under the field of the segment that CONSUMES the position
    // (a session whose modulus changes holds one tape per field, and an input may be read again by a later segment that the
    // wire lived on into: capi.cpp switch_field).  Segments with different moduli reading one such position would each need
    // their own residue: refused.
    auto field_of_position = [&](bool witness, uint32_t position) -> const FieldHost& {
      const FieldHost* found = nullptr;
      for (size_t g = 0; g < n_segments(s); ++g) {
        const Tape& t = seg_backend(s, g).tape();
        const uint8_t want = witness ? TK_WITNESS : TK_INSTANCE;
        bool reads = false;
        for (size_t i = 0; i < t.size() && !reads; ++i) reads = t.kind[i] == want && t.a[i] == position;
        if (!reads) continue;
        const FieldHost& f = seg_backend(s, g).field();
        if (found && memcmp(found->p, f.p, sizeof f.p) != 0)
          throw std::runtime_error(std::string("GPU backend: ") + (witness ? "witness" : "instance") + " value " + std::to_string(position) +
                                   " is wider than zkgpu_elem_bytes and is read under two field characteristics: hand the batch "
                                   "over with zkgpu_set_inputs in values of a width that holds it");
        found = &f;
      }
      return found ? *found : s->backend.field();
    };
    // a value too wide for the buffer is >= p.  Where only the residue matters (mode 0) the residue goes in; at a strict
    // position (0xFF) all-ones does, which the device flags like any other non-canonical strict input; at a position only
    // zero tests read (0x01) all-ones too -- it is >= p and non-zero, which is all those look at.  A position read both by
    // zero tests and by arithmetic (0x02) would need the residue AND the fact that the integer is not zero: refused.
    auto fill = [&](const std::vector<Value>& vals, std::vector<uint8_t>& buf, const std::vector<uint8_t>& modes, bool witness) {
      for (size_t k = 0; k < vals.size(); ++k) {
        bool big = false;
        size_t sig = vals[k].size();
        while (sig > 0 && vals[k][sig - 1] == 0) --sig;
        put_value(vals[k], &buf[k * w], w, sig > w ? field_of_position(witness, (uint32_t)k) : s->backend.field(), &big);
        const uint8_t mode = k < modes.size() ? modes[k] : 0;
        if (big && (mode == 0xFF || mode == 0x01)) memset(&buf[k * w], 0xff, w);
        if (big && mode == 0x03)
          throw std::runtime_error("GPU backend: input value " + std::to_string(k) + " is wider than the field's limbs and its bits are "
                                   "read as they are (and / xor over this field, Evaluator::get: evaluator.rs:750-752,924-933)");
        if (big && mode == 0x02)
          throw std::runtime_error("GPU backend: input value " + std::to_string(k) + " is wider than the field's limbs and is read both by "
                                   "arithmetic gates (which need its residue) and, through copies, by assert_zero / not (which "
                                   "test the unreduced integer, evaluator.rs:900-906,935-938): hand the batch over with "
                                   "zkgpu_set_inputs in full-width values instead");
      }
    };
    std::vector<uint8_t> mi(ni, 0), mw(nw, 0);
    if (ni) zkgpu_input_modes(s, 0, mi.data(), mi.size());
    if (nw) zkgpu_input_modes(s, 1, mw.data(), mw.size());
    fill(li, inst, mi, false);
    fill(lw, wit, mw, true);
    if (!s->prev.empty()) {
      std::vector<Engine*> chain = chain_engines(s);
      for (Engine* e : chain) e->synchronize();
      split_lanes(s, 1);   // (one lane: the first device's chain)
      for (Engine* e : chain) e->set_batch(1);
      chain[0]->upload_inputs(inst.data(), wit.data());
      for (size_t k = 1; k < chain.size(); ++k) chain[k]->use_device_inputs(chain[0]->device_instances(), chain[0]->device_witnesses());
      s->results_fresh = false;
      return;
    }
    split_lanes(s, 1);
    s->engine->set_batch(1);
    s->engine->upload_inputs(inst.data(), wit.data());
    s->results_fresh = false;
  });
}

int zkgpu_set_lane_group(zkgpu_session* s, uint32_t lanes) {
  return guarded(s, [&] {
    s->lane_group = lanes;
    for (Engine* e : configurable_engines(s)) e->set_lane_group(lanes);
  });
}

int zkgpu_set_option(zkgpu_session* s, const char* key, const char* value) {
  return guarded(s, [&] {
    const std::string k = key ? key : "", v = value ? value : "";
    // the scheduler of a streamed ingest took its options when the first window was cut: a later change would be ignored
    // silently by the windows already scheduled -- refuse it instead
    if (s->stream && (k == "fuse" || k == "pair" || k == "fermat" || k == "propagate_copies" || k == "sort_by_operand" ||
                      k == "bank_aware" || k == "strand_width" || k == "strand_lds" || k == "strand_prefetch" || k == "strand_merge" || k == "strand_reassociate" || k == "strand_split_inputs" || k == "bool_narrow_width" || k == "schedule_threads"))
      throw std::runtime_error(k + ": the streamed schedule has started (option \"stream\"); set scheduling options before the first Relation message");
    if (k == "bool_path") {
      if (v == "auto") s->bool_path = 0;
      else if (v == "hbm") s->bool_path = 1;
      else if (v == "lds") s->bool_path = 2;
      else throw std::runtime_error("bool_path must be auto, hbm or lds");
      if (s->engine) { s->engine.reset(); s->engine_loaded = false; }  // re-created with the new choice on the next replay call
    } else if (k == "max_tape_ops") {
      s->backend.set_max_ops(strtoull(v.c_str(), nullptr, 10));
    } else if (k == "streams") {
      s->n_streams = (uint32_t)std::max(1, std::min(4, atoi(v.c_str())));
      for (Engine* e : configurable_engines(s)) e->set_streams(s->n_streams);
    } else if (k == "level_ops_per_wave") {
      s->level_ops_per_wave = (uint32_t)std::max(1, atoi(v.c_str()));
      for (Engine* e : configurable_engines(s)) e->set_level_ops_per_wave(s->level_ops_per_wave);
    } else if (k == "devices") {
      // "0,1,2,3" (HIP device indices; a device may be listed twice: two engines share it) or "" = the current device
      if (s->engine_loaded) throw std::runtime_error("devices must be set before the first zkgpu_set_inputs* call");
      std::vector<int> devs;
      size_t at = 0;
      while (at < v.size()) {
        size_t end = v.find(',', at);
        if (end == std::string::npos) end = v.size();
        const std::string tok = v.substr(at, end - at);
        if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) throw std::runtime_error("devices: a comma-separated list of device indices");
        devs.push_back(atoi(tok.c_str()));
        at = end + 1;
      }
      if (devs.size() > 64) throw std::runtime_error("devices: at most 64 engines");
      s->devices = devs;
    } else if (k == "inspect_segment") {
      s->inspect_segment = v.empty() ? -1 : atoi(v.c_str());
    } else if (k == "force_rccl") {
      s->force_rccl = v != "0";
    } else if (k == "stream") {
      // tape entries per window; "1" = the default window of 131072 entries; before the first Relation message
      if (s->backend.tape().size() || s->finalized) throw std::runtime_error("stream must be set before the first Relation message");
      const long n = atol(v.c_str());
      s->stream_explicit = true;
      s->stream_window = n <= 0 ? 0 : n == 1 ? 131072u : (uint32_t)std::max<long>(n, 16);
      s->backend.set_window(s->stream_window, s->stream_window ? stream_cut : nullptr, s);
    } else if (k == "strand_width") {
      s->strand_width = (uint32_t)std::max(0, atoi(v.c_str()));
    } else if (k == "bank_aware") {
      s->bank_aware = v != "0";
    } else if (k == "strand_lds") {
      s->strand_lds = v != "0";
    } else if (k == "strand_prefetch") {
      s->strand_prefetch = v != "0";
    } else if (k == "strand_merge") {
      s->strand_merge = v != "0";
    } else if (k == "strand_reassociate") {
      s->strand_reassociate = v != "0";
    } else if (k == "strand_split_inputs") {
      s->strand_split_inputs = v != "0";
    } else if (k == "r1cs_coef_classes") {
      if (s->r1cs_ready) throw std::runtime_error("r1cs_coef_classes: set it before the rows are made (zkgpu_r1cs_from_tape / zkgpu_r1cs_load_csr)");
      s->r1cs_coef_classes = v != "0";
    } else if (k == "bool_narrow_width") {
      s->bool_narrow_width = (uint32_t)std::max(3, std::min(2048, atoi(v.c_str())));
    } else if (k == "schedule_threads") {
      s->sched_threads = (uint32_t)std::max(0, atoi(v.c_str()));
    } else if (k == "hot_waves") {
      s->hot_waves = (uint32_t)std::max(0, atoi(v.c_str()));
      for (Engine* e : configurable_engines(s)) e->set_hot_waves(s->hot_waves);
    } else if (k == "graph") {
      if (v == "0" || v == "1") s->graph_mode = atoi(v.c_str());
      else throw std::runtime_error("graph must be 0 or 1");
      for (Engine* e : configurable_engines(s)) e->set_graph_mode(s->graph_mode);
    } else if (k == "xcd_map") {
      s->xcd_map = v != "0";
      for (Engine* e : configurable_engines(s)) e->set_xcd_map(s->xcd_map);
    } else if (k == "fermat") {
      s->fermat = v != "0";
    } else if (k == "pair") {
      s->pair = v != "0";
    } else if (k == "propagate_copies") {
      s->propagate_copies = v != "0";
    } else if (k == "fuse") {
      s->fuse = v != "0";
    } else if (k == "sort_by_operand") {
      s->sort_by_operand = std::max(0, std::min(3, atoi(v.c_str())));
    } else if (k == "validate") {
      if (v == "prover") s->validator.reset(new Validator(Validator::new_as_prover()));
      else if (v == "verifier") s->validator.reset(new Validator(Validator::new_as_verifier()));
      else if (v == "off") s->validator.reset();
      else throw std::runtime_error("validate must be prover, verifier or off");
    } else if (k == "validator_max_steps") {
      if (!s->validator) throw std::runtime_error("validator_max_steps: the validator is not enabled");
      s->validator->set_max_steps(strtoull(v.c_str(), nullptr, 10));
    } else if (k == "metrics") {
      if (v != "0") s->stats.reset(new Stats());
      else s->stats.reset();
    } else {
      throw std::runtime_error("unknown option " + k);
    }
  });
}

size_t zkgpu_modulus(const zkgpu_session* s, uint8_t* buf, size_t cap) {
  if (!s || !s->backend.field_set()) return 0;
  const Value& m = seg_backend(s, inspected(s)).modulus();
  if (buf && cap) memcpy(buf, m.data(), std::min(cap, m.size()));
  return m.size();
}
uint32_t zkgpu_message_values(const zkgpu_session* s, int witness) {
  if (!s) return 0;
  return (uint32_t)(witness ? s->backend.lane0_witnesses().size() : s->backend.lane0_instances().size());
}
size_t zkgpu_message_value(const zkgpu_session* s, int witness, uint32_t index, uint8_t* buf, size_t cap) {
  if (!s) return 0;
  const auto& v = witness ? s->backend.lane0_witnesses() : s->backend.lane0_instances();
  if (index >= v.size()) return 0;
  if (buf && cap) memcpy(buf, v[index].data(), std::min(cap, v[index].size()));
  return v[index].size();
}

size_t zkgpu_validator_violations(zkgpu_session* s, char* buf, size_t cap) {
  if (!s || !s->validator) return copy_out("", buf, cap);
  return copy_out(join_lines(s->validator->get_violations()), buf, cap);
}
int zkgpu_validator_count(zkgpu_session* s) {
  if (!s || !s->validator) return -1;
  return (int)s->validator->get_violations().size();
}
int zkgpu_validator_live_wires(zkgpu_session* s) {
  if (!s || !s->validator) return -1;
  return s->validator->has_live_wires() ? 1 : 0;
}
size_t zkgpu_stats_json(zkgpu_session* s, char* buf, size_t cap) {
  if (!s || !s->stats) return copy_out("", buf, cap);
  return copy_out(s->stats->to_json_pretty(), buf, cap);
}
size_t zkgpu_stats_warnings(zkgpu_session* s, char* buf, size_t cap) {
  if (!s || !s->stats) return copy_out("", buf, cap);
  return copy_out(join_lines(s->stats->warnings), buf, cap);
}

int zkgpu_uses_lds_path(zkgpu_session* s) {
  if (!s) return -1;
  int r = 0;
  if (guarded(s, [&] { need_engine(s); r = s->engine->uses_lds_path() ? 1 : 0; }) != 0) return -1;
  return r;
}

int zkgpu_replay(zkgpu_session* s) {
  return guarded(s, [&] {
    need_engine(s);
    s->results_fresh = false;
    if (!s->prev.empty()) {
      // field segment after field segment on one stream per device; between two, the wires that live on travel as canonical integers
      per_engine(active_engines(s), [&](size_t d) {
        std::vector<Engine*> chain = chain_engines(s, d);
        for (size_t k = 0; k < chain.size(); ++k) {
          chain[k]->replay(false);
          if (k + 1 < chain.size()) {
            const FieldSegment& seg = *s->prev[k];
            std::vector<uint32_t> slots(seg.carried_out.size());
            for (size_t q = 0; q < slots.size(); ++q) slots[q] = seg.sched.slot_of[seg.carried_out[q]];
            chain[k]->carry_out(slots, chain[k + 1]);
          }
        }
      });
    } else if (s->peers.empty()) {
      s->engine->replay(false);
    } else {
      std::vector<Engine*> eng = all_engines(s);
      per_engine(active_engines(s), [&](size_t k) { eng[k]->replay(false); });
    }
  });
}
int zkgpu_replay_timed(zkgpu_session* s) {
  return guarded(s, [&] {
    need_engine(s);
    single_device_only(s, "zkgpu_replay_timed");
    single_segment_only(s, "zkgpu_replay_timed");
    s->results_fresh = false;
    s->engine->replay(true);
    s->engine->synchronize();
  });
}
int zkgpu_synchronize(zkgpu_session* s) {
  return guarded(s, [&] {
    need_engine(s);
    std::vector<Engine*> eng = all_engines(s);
    if (s->peers.empty()) {
      for (auto& seg : s->prev) seg->engine->synchronize();
      s->engine->synchronize();
    } else {
      for (size_t k : active_engines(s)) {
        for (size_t g = 0; g < s->prev.size(); ++g) seg_engine_on(s, g, k)->synchronize();
        eng[k]->synchronize();
      }
    }
  });
}
float zkgpu_last_replay_ms(const zkgpu_session* s) {   // several devices: the slowest share
  if (!s || !s->engine) return 0.f;
  float ms = s->engine->last_replay_ms();
  for (const auto& p : s->peers) ms = std::max(ms, p->last_replay_ms());
  for (const auto& seg : s->prev)
    if (seg->engine) ms += seg->engine->last_replay_ms();   // field segments run one after the other
  return ms;
}

size_t zkgpu_launch_timings(const zkgpu_session* s, float* ms, uint32_t* ops, size_t cap) {
  if (!s || !s->engine) return 0;
  const auto& t = s->engine->launch_timings();
  for (size_t i = 0; i < t.size() && i < cap; ++i) {
    if (ms) ms[i] = t[i].ms;
    if (ops) ops[i] = t[i].count;
  }
  return t.size();
}

int zkgpu_counts(zkgpu_session* s, uint64_t out[2]) {
  return guarded(s, [&] {
    need_engine(s);
    if (s->peers.empty() && !s->force_rccl) s->engine->download(nullptr, nullptr, out);
    else total_counts(s, out);
  });
}
uint64_t zkgpu_rccl_reductions(const zkgpu_session* s) { return s ? s->rccl_reductions : 0; }
size_t zkgpu_rccl_note(const zkgpu_session* s, char* buf, size_t cap) { return s ? copy_out(s->rccl_note, buf, cap) : 0; }
void* zkgpu_counts_device(zkgpu_session* s) { return (s && s->engine && s->peers.empty()) ? s->engine->counts_device() : nullptr; }
void* zkgpu_stream(zkgpu_session* s) { return (s && s->engine && s->peers.empty()) ? s->engine->stream() : nullptr; }
int zkgpu_device_count(void) { return visible_devices(); }
int zkgpu_n_engines(const zkgpu_session* s) { return s ? (int)(s->engine_loaded ? 1 + s->peers.size() : std::max<size_t>(1, s->devices.size())) : 0; }

int zkgpu_lane_results(zkgpu_session* s, uint32_t* first_fail, uint32_t* flags) {
  return guarded(s, [&] {
    fetch_results(s);
    if (first_fail) memcpy(first_fail, s->first_fail.data(), s->first_fail.size() * 4);
    if (flags) memcpy(flags, s->flags.data(), s->flags.size() * 4);
  });
}

size_t zkgpu_lane_violations(zkgpu_session* s, uint32_t lane, char* buf, size_t cap) {
  if (!s) return 0;
  std::vector<std::string> v;
  try {
    fetch_results(s);
    if (lane >= s->first_fail.size()) throw std::runtime_error("lane out of range");
    // evaluator.rs:199-208 for this lane: the first error in execution order wins
    if (s->used_evaluator)
      for (const std::string& m : s->ev.get_violations())
        if (!s->ev.has_error() || m != s->ev.error()) v.push_back(m);  // "Did not receive any gate to verify."
    const uint32_t ff = s->first_fail[lane];
    if (s->flags[lane] & ZKGPU_LANE_NONCANONICAL) {
      v.push_back("GPU backend: an instance or witness value is not canonical (>= field characteristic) and reaches and / xor "
                  "over an odd field, Evaluator::get (a wire alive at the end) or, over GF(2), both a zero test and a gate "
                  "without passing through an arithmetic gate; the reference evaluates those on the unreduced integer "
                  "(evaluator.rs:896-946) and this path does not");
    } else if (ff != ZKGPU_NO_FAIL) {
      size_t g = n_segments(s) - 1;   // the field segment whose asserts include sequence number ff
      while (g > 0 && ff < seg_assert_base(s, g)) --g;
      v.push_back("Wire_" + std::to_string(seg_backend(s, g).tape().assert_wire.at(ff - seg_assert_base(s, g))) +
                  " (may be weighted) should be 0, while it is not");
    } else if (s->ev.has_error()) {
      v.push_back(s->ev.error());
    }
  } catch (const std::exception& e) {
    s->last_error = e.what();
    return 0;
  }
  return copy_out(join_lines(v), buf, cap);
}

int zkgpu_dump_trace_values(zkgpu_session* s, uint64_t first, uint64_t count, uint8_t* out) {
  return guarded(s, [&] {
    need_engine(s);
    if (!s->retain_all) throw std::runtime_error("zkgpu_finalize(retain_all=1) is required for trace dumps");
    need_value_index(s);
    if (s->prev.empty()) {
      if (first + count > s->value_op_index.size()) throw std::runtime_error("trace range out of bounds");
      std::vector<uint32_t> slots(count);
      for (uint64_t k = 0; k < count; ++k) slots[k] = s->sched.slot_of[s->value_op_index[first + k]];
      std::vector<uint8_t> tmp;
      dump_slots_all(s, slots, &tmp);
      if (!tmp.empty()) memcpy(out, tmp.data(), tmp.size());
      return;
    }
    // field segments: the calls run through the segments in order; every value is written in the session's element
    // width (the limbs of the widest field), zero-extended
    if (first + count > zkgpu_tape_value_ops(s)) throw std::runtime_error("trace range out of bounds");
    const uint32_t w = session_elem_bytes(s), batch = s->batch;
    memset(out, 0, (size_t)batch * count * w);
    uint64_t base = 0;
    for (size_t g = 0; g < n_segments(s); ++g) {
      const std::vector<uint32_t>& idx = g < s->prev.size() ? s->prev[g]->value_op_index : s->value_op_index;
      const uint64_t lo = std::max<uint64_t>(first, base), hi = std::min<uint64_t>(first + count, base + idx.size());
      if (lo < hi) {
        std::vector<uint32_t> slots(hi - lo);
        for (uint64_t k = lo; k < hi; ++k) slots[k - lo] = seg_sched(s, g).slot_of[idx[k - base]];
        std::vector<uint8_t> tmp;
        if (s->peers.empty()) {
          seg_engine(s, g)->dump_slots(slots, &tmp);
        } else {   // shares are contiguous and in lane order
          for (size_t d : active_engines(s)) {
            std::vector<uint8_t> part;
            seg_engine_on(s, g, d)->dump_slots(slots, &part);
            tmp.insert(tmp.end(), part.begin(), part.end());
          }
        }
        const uint32_t ws = seg_engine(s, g)->elem_bytes();
        for (uint32_t lane = 0; lane < batch; ++lane)
          for (uint64_t k = lo; k < hi; ++k)
            memcpy(out + ((size_t)lane * count + (k - first)) * w, tmp.data() + ((size_t)lane * (hi - lo) + (k - lo)) * ws, ws);
      }
      base += idx.size();
    }
  });
}

int zkgpu_get_wire(zkgpu_session* s, uint64_t wire_id, uint8_t* out) {
  if (!s) return -1;
  const TapeWire* w = s->ev.get(wire_id);
  if (!w) return 3;  // "No value given for wire_{id}"
  return guarded(s, [&] {
    need_engine(s);
    std::vector<uint8_t> tmp;
    uint32_t ws = s->engine->elem_bytes();
    const auto& raw = s->sched.raw_source;
    const auto it = std::lower_bound(raw.begin(), raw.end(), std::make_pair(w->h, 0u));
    if (it != raw.end() && it->first == w->h) {
      // the wire is an input the relation has only copied: Evaluator::get returns the integer the witness holds, reduced
      // or not (evaluator.rs:750-752, 940-946) -- read it where the caller put it
      const uint32_t q = it->second - 2;
      if ((q & 3) == 3) {   // a constant >= p: the integer the relation wrote, for every lane
        const Value& c = s->backend.tape().consts[s->sched.raw_const_of[q >> 2]];
        size_t n = c.size();
        while (n > 0 && c[n - 1] == 0) --n;
        ws = session_elem_bytes(s);
        if (n > ws) throw std::runtime_error("zkgpu_get_wire: the wire holds a constant wider than zkgpu_elem_bytes");
        tmp.assign((size_t)s->batch * ws, 0);
        for (uint32_t lane = 0; lane < s->batch; ++lane) memcpy(&tmp[(size_t)lane * ws], c.data(), n);
      } else if (s->peers.empty()) {
        s->engine->read_input(q & 3, q >> 2, &tmp, &ws);
      } else {
        std::vector<Engine*> eng = all_engines(s);
        for (size_t k : active_engines(s)) {   // shares are contiguous and in lane order
          std::vector<uint8_t> part;
          eng[k]->read_input(q & 3, q >> 2, &part, &ws);
          tmp.insert(tmp.end(), part.begin(), part.end());
        }
      }
    } else {
      std::vector<uint32_t> slots(1, s->sched.slot_of[w->h]);
      dump_slots_all(s, slots, &tmp);
    }
    const uint32_t we = session_elem_bytes(s);
    if (we == ws) {
      if (!tmp.empty()) memcpy(out, tmp.data(), tmp.size());
    } else {   // field segments: the caller's element width is that of the widest field
      memset(out, 0, (size_t)s->batch * we);
      for (uint32_t lane = 0; lane < s->batch; ++lane) memcpy(out + (size_t)lane * we, tmp.data() + (size_t)lane * ws, std::min(ws, we));
    }
  });
}

int zkgpu_r1cs_from_tape(zkgpu_session* s, int use_correction) {
  return guarded(s, [&] {
    single_segment_only(s, "the R1CS conversion");
    if (!s->backend.field_set()) throw std::runtime_error("no Relation ingested: the field is not set");
    if (s->backend.field().is_two || s->backend.field().generic)
      throw std::runtime_error("R1CS conversion on the GPU path needs an odd field characteristic of at most 512 bits");
    const Tape& t = s->backend.tape();
    Value modulus(4 * kFieldWords, 0);
    for (int i = 0; i < 4 * kFieldWords; ++i) modulus[i] = (uint8_t)(s->backend.field().p[i / 4] >> (8 * (i % 4)));
    s->r1cs = r1cs_from_tape(t, s->backend.field(), modulus, use_correction != 0);
    s->r1cs_ready = true;
    s->r1cs_loaded_csr = false;
    s->r1cs_extra_vars = 0;
    s->r1cs_rows_dev.clear();
    if (s->finalized && s->retain_all && !use_correction) {
      std::vector<uint32_t> slot_of_var(s->r1cs.n_vars, 0xFFFFFFFFu);
      for (size_t i = 0; i < t.size(); ++i)
        if (t.kind[i] != TK_COPY && t.kind[i] != TK_ASSERT && s->r1cs.var_of_op[i] != kNoVar)
          slot_of_var[s->r1cs.var_of_op[i]] = s->sched.slot_of[i];
      build_device_rows(s, [&](uint64_t var) { return var == kVarOne ? 0xFFFFFFFFu : slot_of_var[var]; });
    }
  });
}

int zkgpu_r1cs_info(const zkgpu_session* s, uint64_t out[4]) {
  if (!s || !s->r1cs_ready) return 1;
  out[0] = s->r1cs.n_rows();
  out[1] = s->r1cs.n_vars;
  out[2] = s->r1cs.terms.size();
  out[3] = s->r1cs.coefs.size();
  return 0;
}

int zkgpu_r1cs_class_counts(const zkgpu_session* s, uint64_t out[3]) {
  if (!s || !s->r1cs_ready) return 1;
  out[0] = out[1] = out[2] = 0;
  for (const R1csRowDev& d : s->r1cs_rows_dev) {
    const uint32_t flags = d.counts >> 24;
    for (uint32_t shift : {kR1csClassShiftA, kR1csClassShiftB, kR1csClassShiftC}) ++out[std::min<uint32_t>((flags >> shift) & 3, 2)];
  }
  return 0;
}

int zkgpu_r1cs_export(const zkgpu_session* s, uint32_t* row_ptr, uint64_t* term_var, uint32_t* term_coef,
                      uint64_t* var_of_op) {
  if (!s || !s->r1cs_ready) return 1;
  const R1cs& r = s->r1cs;
  if (row_ptr && !r.row_ptr.empty()) memcpy(row_ptr, r.row_ptr.data(), r.row_ptr.size() * 4);
  for (size_t i = 0; i < r.terms.size(); ++i) {
    if (term_var) term_var[i] = r.terms[i].var;
    if (term_coef) term_coef[i] = r.terms[i].coef;
  }
  if (var_of_op && !r.var_of_op.empty()) memcpy(var_of_op, r.var_of_op.data(), r.var_of_op.size() * 8);
  return 0;
}

size_t zkgpu_r1cs_coef_bytes(const zkgpu_session* s, uint32_t index, uint8_t* out, size_t cap) {
  if (!s || !s->r1cs_ready || index >= s->r1cs.coefs.size()) return 0;
  const Value& v = s->r1cs.coefs[index];
  if (out && cap >= v.size() && !v.empty()) memcpy(out, v.data(), v.size());
  return v.size();
}

int zkgpu_r1cs_load_csr(zkgpu_session* s, uint32_t n_rows, const uint32_t* row_ptr, const uint64_t* term_var,
                        const uint32_t* term_coef, const uint8_t* coef_bytes, uint32_t coef_width, uint32_t n_coefs,
                        uint32_t n_extra_vars) {
  return guarded(s, [&] {
    single_segment_only(s, "zkgpu_r1cs_load_csr");
    if (!s->finalized || !s->retain_all) throw std::runtime_error("zkgpu_finalize(retain_all=1) first: variables are tape values");
    if (s->backend.field().is_two || s->backend.field().generic)
      throw std::runtime_error("R1CS rows on the GPU path need an odd field characteristic of at most 512 bits");
    if (s->engine_loaded) throw std::runtime_error("load the CSR before the first zkgpu_set_inputs* call (the table is sized once)");
    if (!row_ptr || (n_coefs && (!coef_bytes || coef_width == 0)))
      throw std::runtime_error("zkgpu_r1cs_load_csr: row_ptr / coefficient bytes missing or coef_width is 0");
    R1cs r;
    r.row_ptr.assign(row_ptr, row_ptr + 3 * (size_t)n_rows + 1);
    // the term arrays hold row_ptr[3 * n_rows] entries: row_ptr must start at 0 and never step back, or a row
    // would index outside them
    if (r.row_ptr[0] != 0) throw std::runtime_error("zkgpu_r1cs_load_csr: row_ptr[0] must be 0");
    for (size_t i = 1; i < r.row_ptr.size(); ++i)
      if (r.row_ptr[i] < r.row_ptr[i - 1])
        throw std::runtime_error("zkgpu_r1cs_load_csr: row_ptr decreases at entry " + std::to_string(i));
    const size_t n_terms = r.row_ptr.back();
    if (n_terms && (!term_var || !term_coef)) throw std::runtime_error("zkgpu_r1cs_load_csr: term arrays missing");
    r.terms.resize(n_terms);
    for (size_t i = 0; i < n_terms; ++i) {
      if (term_coef[i] >= n_coefs)
        throw std::runtime_error("zkgpu_r1cs_load_csr: term " + std::to_string(i) + " names coefficient " +
                                 std::to_string(term_coef[i]) + " of " + std::to_string(n_coefs));
      r.terms[i] = R1csTerm{term_var[i], term_coef[i]};
    }
    for (uint32_t i = 0; i < n_coefs; ++i)
      r.coefs.emplace_back(coef_bytes + (size_t)i * coef_width, coef_bytes + (size_t)(i + 1) * coef_width);
    need_value_index(s);
    const uint64_t n_ops = s->value_op_index.size();
    r.n_vars = n_ops + n_extra_vars;
    s->r1cs = std::move(r);
    s->r1cs_ready = true;
    s->r1cs_loaded_csr = true;
    s->r1cs_extra_vars = n_extra_vars;
    const uint32_t base = s->sched.n_slots;
    build_device_rows(s, [&](uint64_t var) -> uint32_t {
      if (var == ~0ull) return 0xFFFFFFFFu;
      if (var < n_ops) return s->sched.slot_of[s->value_op_index[var]];
      if (var - n_ops >= n_extra_vars) throw std::runtime_error("R1CS term names a variable out of range");
      return base + (uint32_t)(var - n_ops);
    });
  });
}

int zkgpu_r1cs_assign(zkgpu_session* s, uint32_t first_row, uint32_t n_rows) {
  return guarded(s, [&] {
    r1cs_check_assignable(s, first_row, n_rows);
    r1cs_to_device(s);
    per_active_engine(s, [&](Engine* e) { e->r1cs_run(true, first_row, n_rows); });
  });
}

int zkgpu_r1cs_check(zkgpu_session* s) {
  return guarded(s, [&] {
    r1cs_to_device(s);
    per_active_engine(s, [&](Engine* e) {
      e->r1cs_begin_check();
      e->r1cs_run(false, 0, (uint32_t)s->r1cs_rows_dev.size());
      e->r1cs_finish_check();
    });
  });
}

int zkgpu_r1cs_results(zkgpu_session* s, uint32_t* first_fail_row, uint64_t counts[2]) {
  return guarded(s, [&] {
    need_engine(s);
    std::vector<uint32_t> ff;
    if (s->peers.empty()) {
      s->engine->r1cs_results(first_fail_row ? &ff : nullptr, counts);
    } else {   // shares are contiguous and in lane order; the counts are summed on the host (exact either way)
      std::vector<Engine*> eng = all_engines(s);
      if (counts) counts[0] = counts[1] = 0;
      for (size_t k : active_engines(s)) {
        std::vector<uint32_t> part;
        uint64_t c[2] = {0, 0};
        eng[k]->r1cs_results(first_fail_row ? &part : nullptr, c);
        ff.insert(ff.end(), part.begin(), part.end());
        if (counts) {
          counts[0] += c[0];
          counts[1] += c[1];
        }
      }
    }
    if (first_fail_row && !ff.empty()) memcpy(first_fail_row, ff.data(), ff.size() * 4);
  });
}

int zkgpu_r1cs_get_vars(zkgpu_session* s, const uint64_t* vars, uint32_t n_vars, uint8_t* out) {
  return guarded(s, [&] {
    need_engine(s);
    if (!s->r1cs_ready || !s->r1cs_loaded_csr) throw std::runtime_error("zkgpu_r1cs_get_vars needs a CSR loaded with zkgpu_r1cs_load_csr");
    need_value_index(s);
    const uint64_t n_ops = s->value_op_index.size();
    std::vector<uint32_t> slots(n_vars);
    for (uint32_t k = 0; k < n_vars; ++k) {
      const uint64_t var = vars[k];
      if (var < n_ops) slots[k] = s->sched.slot_of[s->value_op_index[var]];
      else if (var - n_ops < s->r1cs_extra_vars) slots[k] = s->sched.n_slots + (uint32_t)(var - n_ops);
      else throw std::runtime_error("variable out of range");
    }
    std::vector<uint8_t> tmp;
    dump_slots_all(s, slots, &tmp);
    if (!tmp.empty()) memcpy(out, tmp.data(), tmp.size());
  });
}

int zkgpu_r1cs_get_var(zkgpu_session* s, uint64_t var, uint8_t* out) { return zkgpu_r1cs_get_vars(s, &var, 1, out); }

int zkgpu_r1cs_correction_values(zkgpu_session* s, const uint64_t* tape_ops, uint32_t n_ops, uint8_t* out) {
  return guarded(s, [&] {
    need_engine(s);
    single_device_only(s, "zkgpu_r1cs_correction_values");
    if (!s->retain_all) throw std::runtime_error("zkgpu_finalize(retain_all=1) is required: the quotients are computed from the wire values");
    const Tape& t = s->backend.tape();
    const FieldHost& f = s->backend.field();
    if (f.is_two || f.generic) throw std::runtime_error("quotient wires need an odd field characteristic of at most 512 bits");
    const uint32_t one_const = (uint32_t)t.consts.size();   // the literal 1 of `not` = add_constant(a, 1) (to_r1cs.rs:369-371)
    std::vector<uint32_t> calls, const_words((size_t)(t.consts.size() + 1) * f.nwords, 0);
    for (size_t c = 0; c < t.consts.size(); ++c) {
      size_t n = t.consts[c].size();
      while (n > 0 && t.consts[c][n - 1] == 0) --n;
      if (n > 4 * (size_t)f.nwords) continue;   // checked below, only if a listed call uses it
      for (size_t b = 0; b < n; ++b) const_words[c * f.nwords + b / 4] |= (uint32_t)t.consts[c][b] << (8 * (b % 4));
    }
    const_words[(size_t)one_const * f.nwords] = 1;
    for (uint32_t k = 0; k < n_ops; ++k) {
      const uint64_t i = tape_ops[k];
      if (i >= t.size()) throw std::runtime_error("zkgpu_r1cs_correction_values: not a recorded call");
      const uint8_t kind = t.kind[i];
      uint32_t flags = 0, b = 0;
      switch (kind) {
        case TK_ADD: b = s->sched.slot_of[t.b[i]]; break;
        case TK_MUL: b = s->sched.slot_of[t.b[i]]; flags = 1; break;
        case TK_ADDC: b = t.b[i]; flags = 2; break;
        case TK_MULC: b = t.b[i]; flags = 3; break;
        case TK_NOT: b = one_const; flags = 2; break;
        default: throw std::runtime_error("zkgpu_r1cs_correction_values: call " + std::to_string(i) + " is not add / mul / add_constant / mul_constant / not");
      }
      if ((flags & 2) && b < t.consts.size()) {
        size_t n = t.consts[b].size();
        while (n > 0 && t.consts[b][n - 1] == 0) --n;
        if (n > 4 * (size_t)f.nwords) throw std::runtime_error("zkgpu_r1cs_correction_values: a constant wider than the field's limbs");
      }
      calls.insert(calls.end(), {s->sched.slot_of[t.a[i]], b, s->sched.slot_of[i], flags});
    }
    std::vector<uint8_t> tmp;
    s->engine->r1cs_corrections(calls, const_words, &tmp);
    if (!tmp.empty()) memcpy(out, tmp.data(), tmp.size());
  });
}

float zkgpu_r1cs_last_ms(const zkgpu_session* s) {   // several devices: the slowest share
  if (!s || !s->engine) return 0.f;
  float ms = s->engine->last_r1cs_ms();
  for (const auto& p : s->peers) ms = std::max(ms, p->last_r1cs_ms());
  return ms;
}

uint64_t zkgpu_table_bytes(const zkgpu_session* s) {
  if (!s || !s->engine) return 0;
  uint64_t n = s->engine->table_bytes();
  for (const auto& p : s->peers) n += p->table_bytes();
  return n;
}

}  // extern "C"
