#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only: the functions are resolved with dlsym when several devices are driven
#include <dlfcn.h>
#include <string.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "device/args.hpp"
#include "engine.hpp"
#include "lds_program.hpp"

namespace zki {
namespace {

void check(hipError_t e, const char* what) {
  if (e != hipSuccess)
    throw std::runtime_error(std::string("HIP: ") + what + ": " + hipGetErrorString(e) +
                             " (no CPU fallback exists for the replay path)");
}
#define HIP_OK(x) check((x), #x)

static_assert(sizeof(DevOp) == sizeof(zkgpu::TapeOp), "DevOp must match the device TapeOp");
static_assert(sizeof(DevOp2) == sizeof(zkgpu::TapeOp2), "DevOp2 must match the device TapeOp2");
static_assert(sizeof(R1csRowDev) == sizeof(zkgpu::R1csRow) && sizeof(R1csTermDev) == sizeof(zkgpu::R1csTerm),
              "host and device R1CS records must match");
static_assert(sizeof(zkgpu::FieldParams) <= 256, "FieldParams must fit Engine::field_params_");
static_assert(kFieldWords == zkgpu::kGenericMaxWords && kMontWords == zkgpu::kMaxWords, "host and device field widths");

static zkgpu::GenericParams generic_params(const FieldHost& f) {
  zkgpu::GenericParams gp;
  memset(&gp, 0, sizeof gp);
  gp.k = f.kwords;
  gp.nwords = f.nwords;
  memcpy(gp.p, f.p, sizeof gp.p);
  memcpy(gp.mu, f.mu, sizeof gp.mu);
  // a power of two?  (one bit set in the k words)
  uint32_t bits_set = 0, at = 0;
  for (uint32_t i = 0; i < gp.k && i < (uint32_t)zkgpu::kGenericMaxWords; ++i)
    for (uint32_t b = 0; b < 32; ++b)
      if (gp.p[i] >> b & 1) { ++bits_set; at = 32 * i + b; }
  gp.pow2_bits = bits_set == 1 && at >= 2 ? at : 0;   // (2 itself is GF(2), never here)
  return gp;
}

template <typename T>
void dfree(T*& p) {
  if (p) {
    (void)hipFree((void*)p);
    p = nullptr;
  }
}

// one launcher set per field width (kernels_arith.hip)
#define ZK_WIDTHS(X) X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16)
void launch_fused(uint32_t nwords, int cls, dim3 grid, size_t lds_pad, hipStream_t st, const zkgpu::ReplayArgs2& a,
                  const zkgpu::FieldParams& fp) {
  switch (nwords) {
#define X(W) case W: zkgpu::launch_replay_fused_w##W(cls, grid, lds_pad, st, a, fp); break;
    ZK_WIDTHS(X)
#undef X
    default: throw std::runtime_error("Engine: unsupported limb count");
  }
}
void launch_strand(uint32_t nwords, int cls, dim3 grid, hipStream_t st, const zkgpu::ReplayArgs2& a, const uint32_t* level_ptr,
                   uint32_t n_levels, size_t lds_bytes, const zkgpu::FieldParams& fp) {
  switch (nwords) {
#define X(W) case W: zkgpu::launch_replay_strand_w##W(cls, grid, st, a, level_ptr, n_levels, lds_bytes, fp); break;
    ZK_WIDTHS(X)
#undef X
    default: throw std::runtime_error("Engine: unsupported limb count");
  }
}
void launch_plain(uint32_t nwords, bool bitops, dim3 grid, hipStream_t st, const zkgpu::ReplayArgs& a, const zkgpu::FieldParams& fp) {
  switch (nwords) {
#define X(W) case W: zkgpu::launch_replay_w##W(bitops, grid, st, a, fp); break;
    ZK_WIDTHS(X)
#undef X
    default: throw std::runtime_error("Engine: unsupported limb count");
  }
}
static_assert(zki::kR1csClassFull == zkgpu::kR1csClassFull && zki::kR1csClassUnit == zkgpu::kR1csClassUnit &&
              zki::kR1csClassSmall == zkgpu::kR1csClassSmall && zki::kR1csClassShiftA == zkgpu::kR1csClassShiftA &&
              zki::kR1csClassShiftB == zkgpu::kR1csClassShiftB && zki::kR1csClassShiftC == zkgpu::kR1csClassShiftC, "r1cs.hpp and device/args.hpp");
void launch_r1cs(uint32_t nwords, bool assign, bool classes, dim3 grid, hipStream_t st, const zkgpu::R1csArgs& a, const zkgpu::FieldParams& fp) {
  switch (nwords) {
#define X(W) case W: zkgpu::launch_r1cs_w##W(assign, classes, grid, st, a, fp); break;
    ZK_WIDTHS(X)
#undef X
    default: throw std::runtime_error("Engine: unsupported limb count");
  }
}
void launch_dump(uint32_t nwords, dim3 grid, hipStream_t st, const uint4* table, uint32_t n_slots, const uint32_t* slots,
                 uint32_t n_dump, uint32_t batch, uint32_t* out, const zkgpu::FieldParams& fp) {
  switch (nwords) {
#define X(W) case W: zkgpu::launch_dump_w##W(grid, st, table, n_slots, slots, n_dump, batch, out, fp); break;
    ZK_WIDTHS(X)
#undef X
    default: throw std::runtime_error("Engine: unsupported limb count");
  }
}

}  // namespace

void Engine::use_device() const {
  if (device_ >= 0) HIP_OK(hipSetDevice(device_));
}

Engine::Engine(int device) : device_(device) {
  int n = 0;
  HIP_OK(hipGetDeviceCount(&n));
  if (n <= 0) throw std::runtime_error("HIP: no GPU visible (no CPU fallback exists for the replay path)");
  if (device >= n) throw std::runtime_error("Engine: device " + std::to_string(device) + " of " + std::to_string(n) + " visible");
  // "the current device" is a property of the calling THREAD: it is resolved once, here, and every entry point sets it
  // again (use_device), whichever host thread it is called from (the streaming worker, the per-device threads)
  if (device_ < 0) HIP_OK(hipGetDevice(&device_));
  use_device();
  hipStream_t st;
  HIP_OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  stream_ = st;
  hipEvent_t ef;
  HIP_OK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
  ev_fork_ = ef;
  for (int k = 0; k < 3; ++k) {
    hipStream_t side;
    hipEvent_t ej;
    HIP_OK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    side_streams_[k] = side;
    ev_join_[k] = ej;
  }
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  ev_begin_ = e0;
  ev_end_ = e1;
  HIP_OK(hipMalloc(&d_counts_, 16));
  HIP_OK(hipMemset(d_counts_, 0, 16));
}

Engine::~Engine() {
  if (device_ >= 0) (void)hipSetDevice(device_);
  if (graph_exec_) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec_);
  free_batch();
  free_windows();
  dfree(d_consts_);
  dfree(d_level_ptr_);
  dfree(d_counts_);
  dfree(d_lds_ops_);
  dfree(d_lds_ops6_);
  dfree(d_lds_blocks_);
  dfree(d_launches_);
  dfree(d_strict_inst_);
  dfree(d_strict_wit_);
  dfree(d_strict_carry_);
  dfree(d_carry_slots_);
  dfree(d_input_aux_);
  dfree(d_generic_params_);
  dfree(d_stamps_);
  dfree(d_r1cs_rows_);
  dfree(d_r1cs_terms_);
  dfree(d_r1cs_coefs_);
  dfree(d_r1cs_counts_);
  if (ev_r1cs_begin_) (void)hipEventDestroy((hipEvent_t)ev_r1cs_begin_);
  if (ev_r1cs_end_) (void)hipEventDestroy((hipEvent_t)ev_r1cs_end_);
  for (void* e : launch_events_) (void)hipEventDestroy((hipEvent_t)e);
  if (ev_begin_) (void)hipEventDestroy((hipEvent_t)ev_begin_);
  if (ev_end_) (void)hipEventDestroy((hipEvent_t)ev_end_);
  for (int k = 0; k < 2; ++k) {
    if (h_stage_[k]) (void)hipHostFree(h_stage_[k]);
    if (ev_stage_[k]) (void)hipEventDestroy((hipEvent_t)ev_stage_[k]);
  }
  if (ev_upload_) (void)hipEventDestroy((hipEvent_t)ev_upload_);
  for (int k = 0; k < 2; ++k)
    if (ev_set_free_[k]) (void)hipEventDestroy((hipEvent_t)ev_set_free_[k]);
  if (copy_stream_) (void)hipStreamDestroy((hipStream_t)copy_stream_);
  if (ev_fork_) (void)hipEventDestroy((hipEvent_t)ev_fork_);
  for (int k = 0; k < 3; ++k) {
    if (ev_join_[k]) (void)hipEventDestroy((hipEvent_t)ev_join_[k]);
    if (side_streams_[k]) (void)hipStreamDestroy((hipStream_t)side_streams_[k]);
  }
  if (owned_stream_) (void)hipStreamDestroy((hipStream_t)owned_stream_);
  else if (stream_) (void)hipStreamDestroy((hipStream_t)stream_);
}

void Engine::free_batch() {
  dfree(d_table_);
  dfree(d_first_fail_);
  dfree(d_flags_);
  for (int k = 0; k < 2; ++k) {
    dfree(d_inst_own_[k]);
    dfree(d_wit_own_[k]);
    set_read_[k] = false;
  }
  upload_pending_ = false;
  dfree(d_packed_inst_);
  dfree(d_packed_wit_);
  dfree(d_carry_);
  dfree(d_r1cs_fail_);
  d_inst_ = d_wit_ = nullptr;
  batch_ = 0;
  graph_dirty_ = true;  // every pointer a captured replay holds is gone
}

// Every index a kernel will dereference is checked on the host before the program is uploaded: a slot, constant
// or input position out of range must be an exception here, never a memory fault on the GPU.
void Engine::validate_program(const Schedule& s, uint32_t n_instance, uint32_t n_witness, uint32_t n_carry) {
  const uint32_t wpc = s.words_per_const ? s.words_per_const : 1;
  // (arithmetic fields: the pool ends with the raw constants of Schedule::raw_const_of, which only source codes name)
  const uint64_t n_consts = s.const_words.size() / wpc - (s.boolean_path ? 0 : std::min<uint64_t>(s.raw_const_of.size(), s.const_words.size() / wpc));
  auto fail = [](size_t i, const char* what) {
    throw std::runtime_error("Engine: program entry " + std::to_string(i) + " has " + what + " out of range");
  };
  // entries of a strand may name values in the workgroup's LDS (kSlotInLds | k, k below the strand's lds_slots): the
  // strands' entry ranges, sorted by first entry
  std::vector<std::pair<uint64_t, uint64_t>> lds_ranges;   // [first, end) of every strand with LDS slots
  std::vector<uint32_t> lds_counts;
  for (const Launch& L : s.launches)
    if (L.sequential && s.fused && L.lds_slots) {
      lds_ranges.emplace_back(L.first, (uint64_t)L.first + L.count);
      lds_counts.push_back(L.lds_slots);
    }
  auto lds_cap_of = [&](size_t i) -> uint32_t {
    auto it = std::upper_bound(lds_ranges.begin(), lds_ranges.end(), std::pair<uint64_t, uint64_t>((uint64_t)i, ~(uint64_t)0));
    if (it == lds_ranges.begin()) return 0;
    --it;
    return i < it->second ? lds_counts[it - lds_ranges.begin()] : 0;
  };
  auto slot = [&](size_t i, uint32_t v) {
    if (v & kSlotInLds) {
      if ((v & ~kSlotInLds) >= lds_cap_of(i)) fail(i, "an LDS slot of a strand");
      return;
    }
    if (v >= s.n_slots) fail(i, ("a wire-table slot (" + std::to_string(v) + " of " + std::to_string(s.n_slots) + ")").c_str());
  };
  // the unreduced source an assert_zero / not entry names (0 none, 1 a constant, 2 + 4 * position + stream)
  auto source = [&](size_t i, uint32_t code) {
    if (code < 2) return;
    const uint32_t q = code - 2, stream = q & 3;
    if ((q >> 2) >= (stream == 0 ? n_instance : stream == 1 ? n_witness : stream == 2 ? n_carry : (uint32_t)s.raw_const_of.size()))
      fail(i, "the input position of its unreduced source");
  };
  auto check = [&](size_t i, uint32_t kind, uint32_t dst, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1, uint32_t ea,
                   uint32_t eb, uint32_t second, uint32_t dst2, uint32_t c0, uint32_t src) {
    switch (kind) {
      case TK_AND: case TK_XOR:
        // over a field other than GF(2) an operand may name an input instead of a slot (device/args.hpp kOperandIsSource)
        slot(i, dst);
        for (uint32_t ref : {a0, b0}) {
          if (!s.boolean_path && (ref & zkgpu::kOperandIsSource)) {
            if ((ref & ~zkgpu::kOperandIsSource) < 2) fail(i, "the input behind an operand");
            source(i, ref & ~zkgpu::kOperandIsSource);
          } else {
            slot(i, ref);
          }
        }
        if (ea || eb || second) fail(i, "its kind word");
        break;
      case TK_ADD: case TK_MUL:
        slot(i, dst); slot(i, a0); slot(i, b0);
        if (ea) slot(i, a1);
        if (eb) slot(i, b1);
        if (second) { slot(i, dst2); slot(i, c0); }
        break;
      case TK_ADDC: case TK_MULC:
        slot(i, dst); slot(i, a0);
        if (b0 >= n_consts) fail(i, "a constant");
        break;
      case TK_COPY: case TK_NZ: slot(i, dst); slot(i, a0); break;
      case TK_NOT: slot(i, dst); slot(i, a0); source(i, src); break;
      case TK_CONST:
        slot(i, dst);
        if (a0 >= n_consts) fail(i, "a constant");
        break;
      case TK_INSTANCE: slot(i, dst); if (a0 >= n_instance) fail(i, "an instance position"); break;
      case TK_WITNESS: slot(i, dst); if (a0 >= n_witness) fail(i, "a witness position"); break;
      case TK_CARRY: slot(i, dst); if (a0 >= n_carry) fail(i, "a carried value"); break;
      // strands: an input in two halves (device/args.hpp OP_INPUT_RAW / OP_INPUT_CONV) -- the raw words go into an LDS value
      case TK_INPUT_RAW:
        slot(i, dst);
        if (!(dst & zkgpu::kSlotInLds) || a1 > 1 || a0 >= (a1 ? n_witness : n_instance)) fail(i, "an input position or its LDS value");
        break;
      case TK_INPUT_CONV:
        slot(i, dst); slot(i, a0);
        if (!(a0 & zkgpu::kSlotInLds) || a1 > 1 || b0 >= (a1 ? n_witness : n_instance)) fail(i, "an input position or its LDS value");
        break;
      case TK_ASSERT: slot(i, a0); source(i, src); break;
      case TK_NOP: break;
      default: fail(i, "an unknown kind");
    }
  };
  // (a million entries: checked in slices on a few threads, the first complaint wins)
  const size_t total = s.fused ? s.ops2.size() : s.ops.size();
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t parts = total < (1u << 17) ? 1 : std::min<size_t>(8, hw ? hw : 1);
  std::vector<std::string> complaint(parts);
  auto slice = [&](size_t part) {
    try {
      const size_t lo = total * part / parts, hi = total * (part + 1) / parts;
      for (size_t i = lo; i < hi; ++i) {
        if (s.fused) {
          const DevOp2& d = s.ops2[i];
          const uint32_t kind = d.kind & 0xFF, ea = (d.kind >> 8) & 3, eb = (d.kind >> 10) & 3, second = (d.kind >> 12) & 3;
          if ((d.kind >> 14) != 0 || ((ea || eb || second) && kind != TK_ADD && kind != TK_MUL)) fail(i, "its kind word");
          check(i, kind, d.dst, d.a0, d.a1, d.b0, d.b1, ea, eb, second, d.pad0, d.pad1, d.a1);
        } else {
          const DevOp& d = s.ops[i];
          // (the unfused entry keeps the source code in the field its kind leaves unused: assert_zero dst, not b)
          const uint32_t src = s.boolean_path ? 0 : d.kind == TK_ASSERT ? d.dst : d.kind == TK_NOT ? d.b : 0;
          check(i, d.kind, d.kind == TK_ASSERT ? 0 : d.dst, d.a, 0, d.b, 0, 0, 0, 0, 0, 0, src);
        }
      }
    } catch (const std::exception& e) {
      complaint[part] = e.what();
    }
  };
  {
    std::vector<std::thread> pool;
    for (size_t part = 1; part < parts; ++part) pool.emplace_back(slice, part);
    slice(0);
    for (auto& th : pool) th.join();
  }
  for (const std::string& c : complaint)
    if (!c.empty()) throw std::runtime_error(c);
  size_t n_ops = s.fused ? s.ops2.size() : s.ops.size();
  for (const Launch& L : s.launches) {
    if ((uint64_t)L.first + L.count > n_ops) throw std::runtime_error("Engine: a launch reaches past the program");
    if (L.sequential && s.fused) {   // a strand: its level bounds must be a non-decreasing walk over exactly its entries
      const uint32_t nl = L.strand_levels;
      if ((uint64_t)L.level_ptr + nl + 1 > s.strand_level_ptr.size()) throw std::runtime_error("Engine: a strand's level bounds are missing");
      const uint32_t* lp = &s.strand_level_ptr[L.level_ptr];
      for (uint32_t q = 0; q < nl; ++q)
        if (lp[q] > lp[q + 1]) throw std::runtime_error("Engine: a strand's level bounds decrease");
      if (lp[0] != 0 || lp[nl] != L.count) throw std::runtime_error("Engine: a strand's level bounds do not cover its entries");
      if ((uint64_t)L.lds_slots * (((s.words_per_const ? s.words_per_const : 1) + 3) / 4) * 64 * 16 > kStrandLdsBytes)
        throw std::runtime_error("Engine: a strand keeps more values in LDS than fit");
    }
  }
}

void Engine::free_windows() {
  for (void*& p : d_windows_) dfree(p);
  d_windows_.clear();
  window_entries_.clear();
  window_entry_bytes_ = 0;
}

void Engine::upload_window(const void* entries, uint64_t n_entries, size_t entry_bytes) {
  use_device();
  if (!d_windows_.empty() && entry_bytes != window_entry_bytes_) throw std::runtime_error("Engine: program windows of two entry formats");
  window_entry_bytes_ = entry_bytes;
  void* d = nullptr;
  HIP_OK(hipMalloc(&d, std::max<size_t>(n_entries * entry_bytes, 64)));
  if (n_entries) HIP_OK(hipMemcpy(d, entries, n_entries * entry_bytes, hipMemcpyHostToDevice));
  d_windows_.push_back(d);
  window_entries_.push_back(n_entries);
}

void Engine::load_program(const Schedule& s, const FieldHost& f, uint32_t n_instance, uint32_t n_witness, uint32_t n_carry,
                          uint32_t carry_words) {
  use_device();
  free_batch();
  dfree(d_consts_);
  dfree(d_lds_ops_);
  dfree(d_launches_);
  sched_ = s.without_entries();
  boolean_ = s.boolean_path;
  nwords_ = f.nwords;
  elem_bytes_ = boolean_ ? 1 : 4 * f.nwords;
  lanes_per_block_ = boolean_ ? 4096 : 64;
  n_inst_ = n_instance;
  n_wit_ = n_witness;
  n_carry_ = n_carry;
  carry_words_ = carry_words;
  if (boolean_ && n_carry) throw std::runtime_error("Engine: values carried between field segments need an arithmetic field");
  if (!in_stride_set_) in_stride_ = elem_bytes_;
  if (in_stride_ < elem_bytes_ || (!boolean_ && in_stride_ % 4)) throw std::runtime_error("Engine: the input stride is narrower than the field's limbs");
  zkgpu::FieldParams fp;
  memset(&fp, 0, sizeof fp);
  memcpy(fp.p, f.p, sizeof fp.p);
  memcpy(fp.r2, f.r2, sizeof fp.r2);
  memcpy(fp.one, f.one, sizeof fp.one);
  fp.n0inv = f.n0inv;
  fp.nwords = f.nwords;
  // lazily reduced sums of K Montgomery products are below (K * p / R + 1) * p: ceil(K * p / R) conditional
  // subtractions make them canonical.  p / R from the top 64 bits of p, rounded up.
  if (!f.is_two && f.nwords >= 2) {
    const long double rho = ((long double)(((uint64_t)f.p[f.nwords - 1] << 32) | f.p[f.nwords - 2]) + 1.0L) / 18446744073709551616.0L;
    for (int k = 1; k <= 4; ++k) {
      uint32_t r = (uint32_t)(k * rho);
      if ((long double)r < k * rho) ++r;
      fp.dot_rounds[k - 1] = std::min<uint32_t>(std::max<uint32_t>(r, 1), (uint32_t)k);
    }
    const long double a3 = 3 * rho + 1;
    fp.lazy_dot3 = (a3 * rho < 0.999L && a3 * a3 * rho < 0.999L) ? 1 : 0;
  }
  memset(field_params_, 0, sizeof field_params_);
  memcpy(field_params_, &fp, sizeof fp);
  generic_ = f.generic;
  if (generic_) {
    // the any-modulus kernels (device/generic_kernels.hpp): canonical residues, p and Barrett's mu in device memory
    if (s.fused || s.boolean_path) throw std::runtime_error("Engine: the any-modulus kernels replay the unfused program");
    if (f.nwords > (uint32_t)zkgpu::kGenericMaxWords) throw std::runtime_error("Engine: the field characteristic is wider than the any-modulus kernels");
    zkgpu::GenericParams gp = generic_params(f);
    if (!d_generic_params_) HIP_OK(hipMalloc(&d_generic_params_, sizeof gp));
    HIP_OK(hipMemcpy(d_generic_params_, &gp, sizeof gp, hipMemcpyHostToDevice));
    generic_k_words_ = gp.k;
  } else if (!s.boolean_path && f.nwords > (uint32_t)zkgpu::kMaxWords) {
    throw std::runtime_error("Engine: the field characteristic is wider than the Montgomery kernels");
  }
  validate_program(s, n_instance, n_witness, n_carry);
  // GF(2): the LDS-resident kernel runs a program of its own (built below from the schedule): the 16-byte entries of the
  // HBM-table kernel are not uploaded for it
  const bool lds_wanted = s.boolean_path && bool_path_ != 1 && lds_program_fits(s);
  if (lds_wanted) {
    free_windows();
  } else {
    // program entries window by window; windows a streamed ingest has already sent stay where they are
    const size_t eb = s.fused ? sizeof(DevOp2) : sizeof(DevOp);
    const uint8_t* host = s.fused ? (const uint8_t*)s.ops2.data() : (const uint8_t*)s.ops.data();
    const size_t n_win = s.window_first_op.empty() ? 0 : s.window_first_op.size() - 1;
    bool keep = !d_windows_.empty() && d_windows_.size() <= n_win && window_entry_bytes_ == eb;
    for (size_t w = 0; keep && w < d_windows_.size(); ++w)
      keep = window_entries_[w] == s.window_first_op[w + 1] - s.window_first_op[w];
    if (!keep) free_windows();
    for (size_t w = d_windows_.size(); w < n_win; ++w)
      upload_window(host + s.window_first_op[w] * eb, s.window_first_op[w + 1] - s.window_first_op[w], eb);
  }
  dfree(d_level_ptr_);
  HIP_OK(hipMalloc(&d_level_ptr_, std::max<size_t>(s.strand_level_ptr.size() * 4, 64)));
  if (!s.strand_level_ptr.empty())
    HIP_OK(hipMemcpy(d_level_ptr_, s.strand_level_ptr.data(), s.strand_level_ptr.size() * 4, hipMemcpyHostToDevice));
  const size_t cbytes = std::max<size_t>(s.const_words.size() * 4, 64);
  HIP_OK(hipMalloc(&d_consts_, cbytes));
  HIP_OK(hipMemset(d_consts_, 0, cbytes));
  if (!s.const_words.empty())
    HIP_OK(hipMemcpy(d_consts_, s.const_words.data(), s.const_words.size() * 4, hipMemcpyHostToDevice));
  {
    // per input position, the mode the scheduler decided (schedule.cpp track_unreduced_values): 0xFF = a value >= p flags
    // the lane; GF(2): 0x01 = packed as `v != 0` (read by zero tests only).  Padded to whole 16-byte rows of the packing kernel.
    auto upload_mask = [&](void*& d, const std::vector<uint8_t>& modes, uint32_t n) {
      dfree(d);
      std::vector<uint8_t> m(((size_t)n + 31) / 16 * 16, 0);
      for (size_t k = 0; k < modes.size() && k < n; ++k) m[k] = modes[k];
      HIP_OK(hipMalloc(&d, m.size()));
      HIP_OK(hipMemcpy(d, m.data(), m.size(), hipMemcpyHostToDevice));
    };
    upload_mask(d_strict_inst_, s.strict_instance, n_instance);
    upload_mask(d_strict_wit_, s.strict_witness, n_witness);
    upload_mask(d_strict_carry_, s.strict_carry, n_carry);
  }
  // GF(2): if every live wire of a 32-witness slice fits in one CU's LDS, run LDS-resident
  constexpr uint32_t kLdsBytes = 160 * 1024;
  lds_path_ = false;
  if (boolean_ && bool_path_ != 1) {
    if (lds_wanted) {
      // the program of the LDS kernel is host work (lds_program.cpp); here it is only uploaded
      std::vector<uint32_t> sizes;
      for (uint32_t br = 1; br <= (uint32_t)zkgpu::kLdsMaxBlockRows; ++br)
        if (zkgpu::bool_lds_has_block_rows(br)) sizes.push_back(br);
      const char* forced = getenv("ZKGPU_LDS_BLOCK_ROWS");   // tuning experiments (tools/c4_diag.py) and the block-shape test
      const LdsProgram P = build_lds_program(s, sizes, forced ? (uint32_t)atoi(forced) : 0u);
      lds_block_rows_ = P.block_rows;
      const std::vector<zkgpu::LdsOp>& lo = P.ops;
      const std::vector<uint16_t>& lo6 = P.rows;
      const std::vector<uint32_t>& blocks = P.blocks;
      const std::vector<uint32_t>& ln = P.chunks;
      n_lds_chunks_ = (uint32_t)(ln.size() / 4);
      dfree(d_lds_blocks_);
      HIP_OK(hipMalloc(&d_lds_blocks_, std::max<size_t>(blocks.size() * 4, 64)));
      if (!blocks.empty()) HIP_OK(hipMemcpy(d_lds_blocks_, blocks.data(), blocks.size() * 4, hipMemcpyHostToDevice));
      HIP_OK(hipMalloc(&d_lds_ops_, std::max<size_t>(lo.size() * sizeof(zkgpu::LdsOp), 64)));
      if (!lo.empty()) HIP_OK(hipMemcpy(d_lds_ops_, lo.data(), lo.size() * sizeof(zkgpu::LdsOp), hipMemcpyHostToDevice));
      dfree(d_lds_ops6_);
      HIP_OK(hipMalloc(&d_lds_ops6_, std::max<size_t>(lo6.size() * 2, 64)));
      if (!lo6.empty()) HIP_OK(hipMemcpy(d_lds_ops6_, lo6.data(), lo6.size() * 2, hipMemcpyHostToDevice));
      HIP_OK(hipMalloc(&d_launches_, std::max<size_t>(ln.size() * 4, 64)));
      if (!ln.empty()) HIP_OK(hipMemcpy(d_launches_, ln.data(), ln.size() * 4, hipMemcpyHostToDevice));
      HIP_OK(zkgpu::bool_lds_set_max_shared((int)kLdsBytes));
      lds_path_ = true;
    } else if (bool_path_ == 2) {
      throw std::runtime_error("Engine: the relation keeps " + std::to_string(s.n_slots) +
                               " wires alive; that does not fit the LDS-resident GF(2) kernel");
    }
  }
  // values must reach the HBM table when somebody can still ask for them
  lds_writeback_ = s.retain_all;
  loaded_ = true;
}

void Engine::set_batch(uint32_t batch) {
  use_device();
  if (!loaded_) throw std::runtime_error("Engine: load_program() first");
  if (batch == 0) throw std::runtime_error("Engine: empty batch");
  if (batch == batch_) return;
  free_batch();
  batch_ = batch;
  lane_blocks_ = (batch + lanes_per_block_ - 1) / lanes_per_block_;
  const uint64_t rec_bytes = boolean_ ? 64 * 8 : (uint64_t)((nwords_ + 3) / 4) * 64 * 16;
  if (boolean_ && extra_slots_) throw std::runtime_error("Engine: extra slots are only supported for arithmetic fields");
  table_slots_ = sched_.n_slots + extra_slots_;
  table_bytes_ = (uint64_t)lane_blocks_ * table_slots_ * rec_bytes;
  size_t free_b = 0, total_b = 0;
  HIP_OK(hipMemGetInfo(&free_b, &total_b));
  if (table_bytes_ > (uint64_t)(0.92 * (double)free_b))
    throw std::runtime_error("Engine: wire table of " + std::to_string(table_bytes_ >> 20) +
                             " MiB does not fit in free HBM (" + std::to_string(free_b >> 20) + " MiB)");
  HIP_OK(hipMalloc(&d_table_, std::max<uint64_t>(table_bytes_, 64)));
  const size_t padded_lanes = (size_t)lane_blocks_ * lanes_per_block_;
  HIP_OK(hipMalloc(&d_first_fail_, padded_lanes * 4));
  HIP_OK(hipMalloc(&d_flags_, padded_lanes * 4));
  HIP_OK(hipMalloc(&d_r1cs_fail_, padded_lanes * 4));
  if (n_carry_) HIP_OK(hipMalloc(&d_carry_, std::max<size_t>((size_t)batch * n_carry_ * carry_words_ * 4, 64)));
  {   // what the input arms of the arithmetic kernels read, behind one pointer (device/args.hpp InputAux)
    zkgpu::InputAux aux;
    memset(&aux, 0, sizeof aux);
    aux.strict_inst = (const uint8_t*)d_strict_inst_;
    aux.strict_wit = (const uint8_t*)d_strict_wit_;
    aux.strict_carry = (const uint8_t*)d_strict_carry_;
    aux.carry = (const zkgpu::u32*)d_carry_;
    aux.n_carry = n_carry_;
    aux.carry_words = carry_words_;
    aux.in_stride_words = in_stride_ / 4;
    // (the pool: the tape's constants in device form, then the raw ones)
    aux.raw_const_base = (uint32_t)(sched_.const_words.size() / std::max<uint32_t>(sched_.words_per_const, 1) - sched_.raw_const_of.size());
    // developer instrumentation: ZKGPU_STRAND_STAMPS=<file> -- clock stamps of the first levels of the strands of >= 256
    // levels, lane block 0 (the last such strand of a replay wins), written to the file by synchronize()
    if (const char* path = getenv("ZKGPU_STRAND_STAMPS")) {
      if (!d_stamps_) HIP_OK(hipMalloc(&d_stamps_, zkgpu::kStampLevels * 16 * 8));
      HIP_OK(hipMemset(d_stamps_, 0, zkgpu::kStampLevels * 16 * 8));
      aux.stamps = (unsigned long long*)d_stamps_;
      stamps_path_ = path;
    }
    if (!d_input_aux_) HIP_OK(hipMalloc(&d_input_aux_, sizeof aux));
    HIP_OK(hipMemcpy(d_input_aux_, &aux, sizeof aux, hipMemcpyHostToDevice));
  }
  if (boolean_) {
    const size_t words = (size_t)lane_blocks_ * 64;
    HIP_OK(hipMalloc(&d_packed_inst_, std::max<size_t>((size_t)n_inst_ * words * 8, 64)));
    HIP_OK(hipMalloc(&d_packed_wit_, std::max<size_t>((size_t)n_wit_ * words * 8, 64)));
  }
}

void Engine::upload_inputs(const uint8_t* inst, const uint8_t* wit) {
  use_device();
  if (!batch_) throw std::runtime_error("Engine: set_batch() first");
  const size_t ib = (size_t)batch_ * n_inst_ * in_stride_, wb = (size_t)batch_ * n_wit_ * in_stride_;
  if (ib && !inst) throw std::runtime_error("Engine: instance values missing");
  if (wb && !wit) throw std::runtime_error("Engine: witness values missing");
  if (!copy_stream_) {
    hipStream_t cs;
    HIP_OK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    copy_stream_ = cs;
    hipEvent_t e;
    HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ev_upload_ = e;
    for (int k = 0; k < 2; ++k) {
      HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ev_set_free_[k] = e;
    }
  }
  // the other set: whatever replay is queued or running keeps reading the current one
  const int k = own_set_ ^ 1;
  hipStream_t cs = (hipStream_t)copy_stream_;
  if (!d_inst_own_[k]) HIP_OK(hipMalloc(&d_inst_own_[k], std::max<size_t>(ib, 64)));
  if (!d_wit_own_[k]) HIP_OK(hipMalloc(&d_wit_own_[k], std::max<size_t>(wb, 64)));
  if (set_read_[k]) HIP_OK(hipStreamWaitEvent(cs, (hipEvent_t)ev_set_free_[k], 0));  // its last reader has finished
  if (ib) staged_upload(d_inst_own_[k], inst, ib);
  if (wb) staged_upload(d_wit_own_[k], wit, wb);
  HIP_OK(hipEventRecord((hipEvent_t)ev_upload_, cs));
  upload_pending_ = true;
  own_set_ = k;
  if (d_inst_ != d_inst_own_[k] || d_wit_ != d_wit_own_[k]) graph_dirty_ = true;
  d_inst_ = d_inst_own_[k];
  d_wit_ = d_wit_own_[k];
}

// The staging copy is what bounds a hand-over (one core moves ~10 GB/s, the link more): large chunks are copied by
// four threads.
static void parallel_copy(uint8_t* dst, const uint8_t* src, size_t n) {
  constexpr size_t kMinPerThread = 1u << 20;
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t parts = std::min<size_t>(std::min<size_t>(4, hw ? hw : 1), n / kMinPerThread);
  if (parts <= 1) {
    memcpy(dst, src, n);
    return;
  }
  std::vector<std::thread> pool;
  const size_t step = (n / parts + 63) & ~(size_t)63;
  for (size_t t = 1; t < parts; ++t) {
    const size_t lo = t * step, hi = std::min(n, lo + step);
    if (lo < hi) pool.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
  }
  memcpy(dst, src, std::min(n, step));
  for (auto& th : pool) th.join();
}

// Host -> HBM through two pinned staging buffers on the copy stream: the CPU copy of chunk k overlaps the DMA of
// chunk k-1, and the whole upload overlaps the replay that is still reading the other input set.  Returns once the
// caller's memory has been read (the last DMAs may still be in flight; the next replay waits for them on the GPU).
// (A plain hipMemcpy from pageable memory measured ~4 GB/s here.)
void Engine::staged_upload(void* dst, const uint8_t* src, size_t bytes) {
  constexpr size_t kChunk = 16u << 20;
  hipStream_t cs = (hipStream_t)copy_stream_;
  // Page-locked caller memory (hipHostMalloc / hipHostRegister, e.g. a pinned torch tensor) is read by the DMA engine
  // directly: no staging copy.  The call still returns only when the caller's buffer has been read.
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type == hipMemoryTypeHost) {
    HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cs));
    HIP_OK(hipStreamSynchronize(cs));
    return;
  }
  (void)hipGetLastError();  // an unregistered pointer makes the query fail: not an error here
  for (int k = 0; k < 2; ++k) {
    if (!h_stage_[k]) HIP_OK(hipHostMalloc(&h_stage_[k], kChunk, hipHostMallocDefault));
    if (!ev_stage_[k]) {
      hipEvent_t e;
      HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      ev_stage_[k] = e;
    }
  }
  size_t done = 0;
  while (done < bytes) {
    const size_t n = std::min(kChunk, bytes - done);
    const int k = stage_next_;
    if (stage_used_[k]) HIP_OK(hipEventSynchronize((hipEvent_t)ev_stage_[k]));  // staging buffer k free again
    parallel_copy((uint8_t*)h_stage_[k], src + done, n);
    HIP_OK(hipMemcpyAsync((uint8_t*)dst + done, h_stage_[k], n, hipMemcpyHostToDevice, cs));
    HIP_OK(hipEventRecord((hipEvent_t)ev_stage_[k], cs));
    stage_used_[k] = true;
    done += n;
    stage_next_ ^= 1;
  }
}

void Engine::use_device_inputs(const void* d_inst, const void* d_wit) {
  if (!batch_) throw std::runtime_error("Engine: set_batch() first");
  if (d_inst_ != d_inst || d_wit_ != d_wit) graph_dirty_ = true;
  d_inst_ = d_inst;
  d_wit_ = d_wit;
}

void Engine::launch_one(size_t li, uint32_t lb0, uint32_t lbs, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  zkgpu::FieldParams fp;
  memcpy(&fp, field_params_, sizeof fp);
  const Launch& L = sched_.launches[li];
  // entries of the launch: in the device buffer of its tape window
  const uint64_t in_window = L.first - sched_.window_first_op[L.window];
  const zkgpu::TapeOp* ops1 = (const zkgpu::TapeOp*)d_windows_[L.window] + in_window;
  const zkgpu::TapeOp2* ops2 = (const zkgpu::TapeOp2*)d_windows_[L.window] + in_window;
  // Wide levels of the fused program walk two ops per wave: half as many waves to dispatch, same registers
  // (measured on C2: 9.83 -> 9.48 ms; three or more per wave lose again to the shorter grid).
  const uint32_t opw = (sched_.fused && !boolean_ && !L.sequential && L.count >= 1024 * level_ops_per_wave_) ? level_ops_per_wave_ : L.ops_per_wave;
  const uint32_t waves = (L.count + opw - 1) / opw;
  const uint32_t chunks = (waves + 3) / 4;
  dim3 grid(chunks, lbs);
  // XCD-aware grid (device/replay_kernels.hpp block_coords): each XCD sweeps whole levels of its own lane blocks
  const uint32_t xcd_chunks = (!boolean_ && xcd_map_ && !L.sequential && lbs % 8 == 0 && chunks >= 8) ? chunks : 0;
  if (xcd_chunks) grid = dim3(chunks * lbs);
  if (boolean_) {
    zkgpu::BoolReplayArgs a;
    memset(&a, 0, sizeof a);
    a.ops = ops1;
    a.n_ops = L.count;
    a.ops_per_wave = L.ops_per_wave;
    a.table = (zkgpu::u64*)d_table_;
    a.n_slots = sched_.n_slots;
    a.batch = batch_;
    a.lb_base = lb0;
    a.total_words = lane_blocks_ * 64;
    a.consts = (const zkgpu::u32*)d_consts_;
    a.packed_inst = (const zkgpu::u64*)d_packed_inst_;
    a.packed_wit = (const zkgpu::u64*)d_packed_wit_;
    a.first_fail = (zkgpu::u32*)verdict_first_fail();
    zkgpu::launch_bool_replay(grid, st, a);
    return;
  }
  if (sched_.fused) {
    zkgpu::ReplayArgs2 a;
    memset(&a, 0, sizeof a);
    a.ops = ops2;
    a.n_ops = L.count;
    a.ops_per_wave = opw;
    a.table = (uint4*)d_table_;
    a.n_slots = table_slots_;
    a.batch = batch_;
    a.lb_base = lb0;
    a.consts = (const zkgpu::u32*)d_consts_;
    a.inst = (const uint8_t*)d_inst_;
    a.wit = (const uint8_t*)d_wit_;
    a.n_inst = n_inst_;
    a.n_wit = n_wit_;
    a.first_fail = (zkgpu::u32*)verdict_first_fail();
    a.lane_flags = (zkgpu::u32*)verdict_flags();
    a.aux = (const zkgpu::InputAux*)d_input_aux_;
    if (L.sequential) {
      // a strand: one workgroup per lane block walks the levels of the run, barrier between levels
      // (a strand has no use for the grid fields: op_stride tells it whether the input buffers hold values of exactly N
      // words -- what the two halves of a split input entry need to know without a trip to InputAux)
      a.xcd_chunks = 0;
      a.op_stride = in_stride_ / 4 == nwords_ ? 1u : 0u;
      launch_strand(nwords_, L.has_bitops ? zkgpu::kFusedAll : zkgpu::kFusedMisc, dim3(lbs), st, a,
                    (const uint32_t*)d_level_ptr_ + L.level_ptr, L.strand_levels, (size_t)L.lds_slots * ((nwords_ + 3) / 4) * 64 * 16, fp);
      return;
    }
    // a level: its Add/Mul entries (scheduled first) run in the instantiation that holds nothing else; the
    // remaining kinds, if any, in a second launch of the general one.  Both read only earlier levels.
    auto part = [&](uint32_t first, uint32_t count, int cls, bool wide) {
      if (!count) return;
      const uint32_t w = (wide && count >= 1024 * level_ops_per_wave_) ? level_ops_per_wave_ : 1;
      const uint32_t nchunks = ((count + w - 1) / w + 3) / 4;
      a.ops = ops2 + first;
      a.n_ops = count;
      a.ops_per_wave = w;
      a.op_stride = w == 1 ? 1 : 4;
      a.xcd_chunks = (xcd_map_ && lbs % 8 == 0 && nchunks >= 8) ? nchunks : 0;
      // hot_waves_: cap on resident waves per SIMD for the Add/Mul kernel (a 256-thread workgroup is one wave per SIMD)
      const size_t pad = (cls == zkgpu::kFusedHot && hot_waves_ >= 3 && hot_waves_ < 8) ? (160 * 1024 / hot_waves_) & ~(size_t)255 : 0;
      launch_fused(nwords_, cls, a.xcd_chunks ? dim3(nchunks * lbs) : dim3(nchunks, lbs), pad, st, a, fp);
    };
    part(0, L.hot_count, zkgpu::kFusedHot, true);
    part(L.hot_count, L.count - L.hot_count, L.has_bitops ? zkgpu::kFusedAll : zkgpu::kFusedMisc, false);
    return;
  }
  zkgpu::ReplayArgs a;
  memset(&a, 0, sizeof a);
  a.ops = ops1;
  a.n_ops = L.count;
  a.ops_per_wave = L.ops_per_wave;
  a.table = (uint4*)d_table_;
  a.n_slots = table_slots_;
  a.batch = batch_;
  a.lb_base = lb0;
  a.consts = (const zkgpu::u32*)d_consts_;
  a.inst = (const uint8_t*)d_inst_;
  a.wit = (const uint8_t*)d_wit_;
  a.n_inst = n_inst_;
  a.n_wit = n_wit_;
  a.first_fail = (zkgpu::u32*)verdict_first_fail();
  a.lane_flags = (zkgpu::u32*)verdict_flags();
  a.aux = (const zkgpu::InputAux*)d_input_aux_;
  a.xcd_chunks = xcd_chunks;
  if (generic_) zkgpu::launch_replay_generic(grid, st, a, (const zkgpu::GenericParams*)d_generic_params_, nwords_, generic_k_words_);
  else launch_plain(nwords_, sched_.has_bitops, grid, st, a, fp);
}

void Engine::launch_range(uint32_t lb0, uint32_t lbs, bool time_each) {
  hipStream_t st = (hipStream_t)stream_;
  // lane blocks split into `parts` contiguous shares, one per stream; the launches of the shares are
  // issued interleaved so that every queue always has the next level ready
  uint32_t parts = time_each ? 1 : std::min<uint32_t>(n_streams_, lbs);
  if (parts < 1) parts = 1;
  uint32_t begin[kMaxStreams + 1];
  if (xcd_map_ && !boolean_ && lbs % 8 == 0) {
    // shares of whole XCD rounds (8 lane blocks), so that every share can use the XCD-aware grid
    const uint32_t rounds = lbs / 8;
    parts = std::min(parts, rounds);
    for (uint32_t p = 0; p <= parts; ++p) begin[p] = lb0 + 8 * (uint32_t)((uint64_t)rounds * p / parts);
  } else {
    for (uint32_t p = 0; p <= parts; ++p) begin[p] = lb0 + (uint32_t)((uint64_t)lbs * p / parts);
  }
  if (parts > 1) {  // fork: the side streams start after everything already queued on the main one
    HIP_OK(hipEventRecord((hipEvent_t)ev_fork_, st));
    for (uint32_t p = 1; p < parts; ++p)
      HIP_OK(hipStreamWaitEvent((hipStream_t)side_streams_[p - 1], (hipEvent_t)ev_fork_, 0));
  }
  for (size_t li = 0; li < sched_.launches.size(); ++li) {
    if (time_each) HIP_OK(hipEventRecord((hipEvent_t)launch_events_[2 * li], st));
    for (uint32_t p = 0; p < parts; ++p)
      launch_one(li, begin[p], begin[p + 1] - begin[p], p == 0 ? stream_ : side_streams_[p - 1]);
    if (time_each) HIP_OK(hipEventRecord((hipEvent_t)launch_events_[2 * li + 1], st));
  }
  for (uint32_t p = 1; p < parts; ++p) {  // join
    HIP_OK(hipEventRecord((hipEvent_t)ev_join_[p - 1], (hipStream_t)side_streams_[p - 1]));
    HIP_OK(hipStreamWaitEvent(st, (hipEvent_t)ev_join_[p - 1], 0));
  }
  HIP_OK(hipGetLastError());
}

void Engine::replay(bool time_each_launch) {
  use_device();
  if (!batch_) throw std::runtime_error("Engine: set_batch() first");
  if ((n_inst_ && !d_inst_) || (n_wit_ && !d_wit_)) throw std::runtime_error("Engine: inputs not set");
  hipStream_t st = (hipStream_t)stream_;
  if (time_each_launch) {
    while (launch_events_.size() < 2 * sched_.launches.size()) {
      hipEvent_t e;
      HIP_OK(hipEventCreate(&e));
      launch_events_.push_back(e);
    }
  }
  if (upload_pending_) {  // the inputs this replay reads are still arriving on the copy stream
    HIP_OK(hipStreamWaitEvent(st, (hipEvent_t)ev_upload_, 0));
    upload_pending_ = false;
  }
  HIP_OK(hipEventRecord((hipEvent_t)ev_begin_, st));
  const bool graphed = use_graph() && !time_each_launch;
  if (graphed) {
    // the memsets, the kernels of every lane group on every stream and the verdict kernel as one captured submission
    if (!graph_exec_ || graph_dirty_) capture_graph();
    HIP_OK(hipGraphLaunch((hipGraphExec_t)graph_exec_, st));
  } else {
    enqueue_replay(time_each_launch);
  }
  HIP_OK(hipEventRecord((hipEvent_t)ev_end_, st));
  if (copy_stream_ && d_inst_ == d_inst_own_[own_set_] && d_wit_ == d_wit_own_[own_set_]) {
    HIP_OK(hipEventRecord((hipEvent_t)ev_set_free_[own_set_], st));  // the next upload into this set waits for it
    set_read_[own_set_] = true;
  }
  HIP_OK(hipGetLastError());
  timings_.clear();
  if (time_each_launch && !lds_path_) {
    HIP_OK(hipStreamSynchronize(st));
    for (size_t li = 0; li < sched_.launches.size(); ++li) {
      LaunchTiming t;
      t.launch = (uint32_t)li;
      t.count = sched_.launches[li].count;
      HIP_OK(hipEventElapsedTime(&t.ms, (hipEvent_t)launch_events_[2 * li], (hipEvent_t)launch_events_[2 * li + 1]));
      timings_.push_back(t);
    }
  }
}

// hipGraph replay is off by default: measured on ROCm 7.2 / MI355X it is slower than issuing the launches on the two
// streams (BN254 Switch example, 18 kernels per replay: 0.113 -> 0.186 ms at 1024 witnesses; C2, 518 kernels per replay:
// 8.83 -> 12.71 ms) -- the captured fork/join of the two lane shares no longer overlaps.  "graph" = 1 keeps the path
// testable for later ROCm releases.
bool Engine::use_graph() const { return graph_mode_ == 1 && !lds_path_; }

void Engine::capture_graph() {
  hipStream_t st = (hipStream_t)stream_;
  if (graph_exec_) {
    (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec_);
    graph_exec_ = nullptr;
  }
  hipGraph_t graph = nullptr;
  HIP_OK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  try {
    enqueue_replay(false);
  } catch (...) {
    (void)hipStreamEndCapture(st, &graph);
    if (graph) (void)hipGraphDestroy(graph);
    throw;
  }
  HIP_OK(hipStreamEndCapture(st, &graph));
  hipGraphExec_t exec = nullptr;
  const hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  HIP_OK(e);
  graph_exec_ = exec;
  graph_dirty_ = false;
}

// The device work of one replay, enqueued on the engine's streams (directly, or into a graph capture).
void Engine::enqueue_replay(bool time_each_launch) {
  hipStream_t st = (hipStream_t)stream_;
  const size_t padded_lanes = (size_t)lane_blocks_ * lanes_per_block_;
  if (chain_first_) {   // (a later field segment of a session adds to the verdict words of the first)
    HIP_OK(hipMemsetAsync(verdict_first_fail(), 0xFF, padded_lanes * 4, st));
    HIP_OK(hipMemsetAsync(verdict_flags(), 0, padded_lanes * 4, st));
    HIP_OK(hipMemsetAsync(verdict_counts(), 0, 16, st));
  }
  if (boolean_) {
    const uint32_t words = lane_blocks_ * 64;
    // both streams in one launch (blockIdx.z)
    zkgpu::launch_pack_inputs(st, (const uint8_t*)d_inst_, n_inst_, (const uint8_t*)d_strict_inst_, (zkgpu::u64*)d_packed_inst_,
                              (const uint8_t*)d_wit_, n_wit_, (const uint8_t*)d_strict_wit_, (zkgpu::u64*)d_packed_wit_, batch_,
                              words, (zkgpu::u32*)verdict_flags());
  }
  if (lds_path_) {
    zkgpu::BoolLdsArgs a;
    memset(&a, 0, sizeof a);
    a.ops = (const zkgpu::LdsOp*)d_lds_ops_;
    a.ops6 = (const zkgpu::u32*)d_lds_ops6_;
    a.blocks = (const zkgpu::u32*)d_lds_blocks_;
    a.block_rows = lds_block_rows_;
    a.chunks = (const zkgpu::u32*)d_launches_;
    a.n_chunks = n_lds_chunks_;
    a.n_slots = sched_.n_slots + zkgpu::kLdsExtraSlots;  // + scratch slots of the padding ops (one per bank), ZERO, ONES
    a.batch = batch_;
    a.n_cols = (batch_ + 31) / 32;
    a.total_words64 = lane_blocks_ * 64;
    a.consts = (const zkgpu::u32*)d_consts_;
    a.packed_inst = (const zkgpu::u32*)d_packed_inst_;
    a.packed_wit = (const zkgpu::u32*)d_packed_wit_;
    a.first_fail = (zkgpu::u32*)verdict_first_fail();
    a.table = (zkgpu::u64*)d_table_;
    a.writeback = (lds_writeback_ || force_writeback_) ? 1 : 0;
    const size_t lds_bytes = (((size_t)sched_.n_slots + zkgpu::kLdsExtraSlots) * 4 + 15) / 16 * 16;
    zkgpu::launch_bool_lds(a.n_cols, lds_bytes, st, a);
  }
  uint32_t group_blocks = lane_blocks_;
  if (lane_group_) {
    group_blocks = std::max<uint32_t>(1, std::min(lane_blocks_, lane_group_ / lanes_per_block_));
  } else if (xcd_map_ && !boolean_ && !lds_path_) {
    // Automatic lane groups: the XCD-aware grid pays off while the wire table of the lanes in flight stays in
    // the 256 MiB Infinity Cache (measured on C2: 102 G gate-ops/s at any batch with 1024-lane groups, against
    // 85-92 G when 2048 or more lanes are replayed at once).  A group is a whole number of XCD rounds per stream.
    const uint64_t per_block = table_bytes_ / std::max<uint32_t>(lane_blocks_, 1);
    const uint32_t unit = 8 * n_streams_;
    const uint64_t fit = per_block ? kInfinityCacheBudget / per_block : lane_blocks_;
    if (lane_blocks_ > unit && fit >= unit) group_blocks = (uint32_t)std::min<uint64_t>(lane_blocks_, fit / unit * unit);
  }
  if (time_each_launch) group_blocks = lane_blocks_;  // per-launch events describe whole-batch launches
  for (uint32_t lb0 = 0; lb0 < lane_blocks_ && !lds_path_; lb0 += group_blocks)
    launch_range(lb0, std::min(group_blocks, lane_blocks_ - lb0), time_each_launch);
  if (chain_last_)
    zkgpu::launch_verdict(dim3((batch_ + 255) / 256), st, (const zkgpu::u32*)verdict_first_fail(), (const zkgpu::u32*)verdict_flags(), batch_,
                          (unsigned long long*)verdict_counts());
}

void Engine::reserve_extra_slots(uint32_t n) {
  if (batch_) throw std::runtime_error("Engine: reserve_extra_slots() must precede set_batch()");
  extra_slots_ = n;
}

int Engine::generic_selftest(const FieldHost& f, int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
  if (!f.generic) return 1;
  const zkgpu::GenericParams gp = generic_params(f);
  return zkgpu::generic_selftest(&gp, op, a, b, out);
}

// (R1CS sessions are refused for fields of the any-modulus path in capi.cpp: the row kernels are Montgomery kernels)
void Engine::r1cs_upload(const std::vector<R1csRowDev>& rows, const std::vector<R1csTermDev>& terms,
                         const std::vector<uint32_t>& coef_words) {
  use_device();
  if (boolean_) throw std::runtime_error("Engine: the R1CS row kernel needs an arithmetic field");
  // host check of every index the row kernel dereferences (same rule as validate_program)
  const uint64_t n_coefs = nwords_ ? coef_words.size() / nwords_ : 0;
  const uint32_t n_table_slots = sched_.n_slots + extra_slots_;
  for (size_t r = 0; r < rows.size(); ++r) {
    const uint32_t n = (rows[r].counts & 0xFF) + ((rows[r].counts >> 8) & 0xFF) + ((rows[r].counts >> 16) & 0xFF);
    if ((uint64_t)rows[r].first + n > terms.size()) throw std::runtime_error("Engine: R1CS row " + std::to_string(r) + " reaches past the term list");
  }
  r1cs_classes_ = false;
  for (size_t r = 0; r < rows.size(); ++r) {
    const uint32_t cnt[3] = {rows[r].counts & 0xFF, (rows[r].counts >> 8) & 0xFF, (rows[r].counts >> 16) & 0xFF};
    const uint32_t flags = rows[r].counts >> 24;
    const uint32_t cls[3] = {(flags >> zkgpu::kR1csClassShiftA) & 3, (flags >> zkgpu::kR1csClassShiftB) & 3, (flags >> zkgpu::kR1csClassShiftC) & 3};
    uint32_t t = rows[r].first;
    for (int part = 0; part < 3; ++part)
      for (uint32_t k = 0; k < cnt[part]; ++k, ++t) {
        if (terms[t].slot != 0xFFFFFFFFu && terms[t].slot >= n_table_slots)
          throw std::runtime_error("Engine: R1CS term " + std::to_string(t) + " names a wire-table slot out of range");
        const uint32_t c = terms[t].coef;
        if (cls[part] == zkgpu::kR1csClassFull) {
          if (c != 0xFFFFFFFFu && c >= n_coefs) throw std::runtime_error("Engine: R1CS term " + std::to_string(t) + " names a coefficient out of range");
        } else {
          r1cs_classes_ = true;
          const uint32_t mag = c & 0x7FFFFFFFu;
          if (cls[part] > zkgpu::kR1csClassSmall || mag == 0 || (cls[part] == zkgpu::kR1csClassUnit && mag != 1))
            throw std::runtime_error("Engine: R1CS term " + std::to_string(t) + " does not fit the coefficient class of its combination");
        }
      }
  }
  dfree(d_r1cs_rows_);
  dfree(d_r1cs_terms_);
  dfree(d_r1cs_coefs_);
  HIP_OK(hipMalloc(&d_r1cs_rows_, std::max<size_t>(rows.size() * sizeof(R1csRowDev), 64)));
  HIP_OK(hipMalloc(&d_r1cs_terms_, std::max<size_t>(terms.size() * sizeof(R1csTermDev), 64)));
  // the pool ends with the Montgomery form of 1: the coefficient of a `1 * w` term inside a chunk that also holds
  // real coefficients (R1csArgs::one_coef)
  zkgpu::FieldParams fpar;
  memcpy(&fpar, field_params_, sizeof fpar);
  std::vector<uint32_t> pool(coef_words);
  pool.insert(pool.end(), fpar.one, fpar.one + nwords_);
  r1cs_one_coef_ = (uint32_t)n_coefs;
  // ... and behind it the Montgomery forms of 2^64 and 2^128 (R mod p doubled 64 and 128 times): what a product of
  // small-class sums is multiplied by when it is stored (r1cs_row_kernel, ASSIGN)
  {
    std::vector<uint32_t> x(fpar.one, fpar.one + nwords_);
    auto twice = [&]() {
      uint64_t c = 0;
      for (uint32_t i = 0; i < nwords_; ++i) {
        const uint64_t y = 2ull * x[i] + c;
        x[i] = (uint32_t)y;
        c = y >> 32;
      }
      bool ge = c != 0;
      if (!ge) {
        ge = true;
        for (uint32_t i = nwords_; i-- > 0;)
          if (x[i] != fpar.p[i]) { ge = x[i] > fpar.p[i]; break; }
      }
      if (ge) {
        uint64_t b = 0;
        for (uint32_t i = 0; i < nwords_; ++i) {
          const uint64_t y = (uint64_t)x[i] - fpar.p[i] - b;
          x[i] = (uint32_t)y;
          b = (y >> 63) & 1;
        }
      }
    };
    for (int k = 0; k < 2; ++k) {
      for (int i = 0; i < 64; ++i) twice();
      pool.insert(pool.end(), x.begin(), x.end());
    }
  }
  HIP_OK(hipMalloc(&d_r1cs_coefs_, std::max<size_t>(pool.size() * 4, 64)));
  if (!rows.empty()) HIP_OK(hipMemcpy(d_r1cs_rows_, rows.data(), rows.size() * sizeof(R1csRowDev), hipMemcpyHostToDevice));
  if (!terms.empty()) HIP_OK(hipMemcpy(d_r1cs_terms_, terms.data(), terms.size() * sizeof(R1csTermDev), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_r1cs_coefs_, pool.data(), pool.size() * 4, hipMemcpyHostToDevice));
  r1cs_rows_ = (uint32_t)rows.size();
  if (!d_r1cs_counts_) HIP_OK(hipMalloc(&d_r1cs_counts_, 16));
  if (!ev_r1cs_begin_) {
    hipEvent_t a, b;
    HIP_OK(hipEventCreate(&a));
    HIP_OK(hipEventCreate(&b));
    ev_r1cs_begin_ = a;
    ev_r1cs_end_ = b;
  }
}

void Engine::r1cs_begin_check() {
  use_device();
  if (!batch_ || !d_r1cs_rows_) throw std::runtime_error("Engine: R1CS rows or batch not set");
  hipStream_t st = (hipStream_t)stream_;
  HIP_OK(hipEventRecord((hipEvent_t)ev_r1cs_begin_, st));
  HIP_OK(hipMemsetAsync(d_r1cs_fail_, 0xFF, (size_t)lane_blocks_ * lanes_per_block_ * 4, st));
  HIP_OK(hipMemsetAsync(d_r1cs_counts_, 0, 16, st));
}

void Engine::r1cs_run(bool assign, uint32_t first_row, uint32_t n_rows) {
  use_device();
  if (!batch_ || !d_r1cs_rows_) throw std::runtime_error("Engine: R1CS rows or batch not set");
  if ((uint64_t)first_row + n_rows > r1cs_rows_) throw std::runtime_error("Engine: R1CS row range out of bounds");
  if (!n_rows) return;
  hipStream_t st = (hipStream_t)stream_;
  zkgpu::FieldParams fp;
  memcpy(&fp, field_params_, sizeof fp);
  zkgpu::R1csArgs a;
  memset(&a, 0, sizeof a);
  a.rows = (const zkgpu::R1csRow*)d_r1cs_rows_;
  a.terms = (const zkgpu::R1csTerm*)d_r1cs_terms_;
  a.coefs = (const zkgpu::u32*)d_r1cs_coefs_;
  a.first_row = first_row;
  a.n_rows = n_rows;
  a.table = (const uint4*)d_table_;
  a.table_out = (uint4*)d_table_;
  a.n_slots = table_slots_;
  a.batch = batch_;
  a.first_fail = (zkgpu::u32*)d_r1cs_fail_;
  a.one_coef = r1cs_one_coef_;
  const dim3 grid((n_rows + 3) / 4, lane_blocks_);
  launch_r1cs(nwords_, assign, r1cs_classes_, grid, st, a, fp);
  HIP_OK(hipGetLastError());
}

void Engine::r1cs_corrections(const std::vector<uint32_t>& calls4, const std::vector<uint32_t>& const_words, std::vector<uint8_t>* out) {
  use_device();
  if (boolean_) throw std::runtime_error("Engine: quotient wires need an arithmetic field");
  if (!batch_ || !d_table_) throw std::runtime_error("Engine: set_batch() and a replay first");
  synchronize();
  const uint32_t n = (uint32_t)(calls4.size() / 4);
  out->assign((size_t)batch_ * n * elem_bytes_, 0);
  if (!n) return;
  const uint64_t n_consts = nwords_ ? const_words.size() / nwords_ : 0;
  for (uint32_t k = 0; k < n; ++k) {   // host check of every index the kernel dereferences
    const uint32_t* c = &calls4[4 * k];
    if (c[0] >= table_slots_ || c[2] >= table_slots_ || ((c[3] & zkgpu::kCorrConstB) ? c[1] >= n_consts : c[1] >= table_slots_))
      throw std::runtime_error("Engine: quotient call " + std::to_string(k) + " names a slot or constant out of range");
  }
  zkgpu::FieldParams fp;
  memcpy(&fp, field_params_, sizeof fp);
  zkgpu::R1csCorrArgs a;
  memset(&a, 0, sizeof a);
  // p^{-1} mod 2^(32 * nwords) by Hensel lifting from -n0inv = p^{-1} mod 2^32: x <- x * (2 - p * x)
  {
    const uint32_t N = nwords_;
    std::vector<uint32_t> x(N, 0), t(N), u(N);
    x[0] = 0u - fp.n0inv;
    auto mul_low = [&](const uint32_t* p, const uint32_t* q, uint32_t* r) {
      std::vector<uint64_t> acc(N + 1, 0);
      for (uint32_t i = 0; i < N; ++i) {
        uint64_t carry = 0;
        for (uint32_t j = 0; i + j < N; ++j) {
          const uint64_t v = (uint64_t)p[i] * q[j] + (acc[i + j] & 0xFFFFFFFFu) + carry;
          acc[i + j] = v & 0xFFFFFFFFu;
          carry = v >> 32;
        }
      }
      for (uint32_t i = 0; i < N; ++i) r[i] = (uint32_t)acc[i];
    };
    for (uint32_t bits = 32; bits < 32 * N; bits *= 2) {
      mul_low(fp.p, x.data(), t.data());            // p * x
      uint64_t borrow = 0;                          // 2 - p * x
      for (uint32_t i = 0; i < N; ++i) {
        const uint64_t d = (uint64_t)(i == 0 ? 2u : 0u) - t[i] - borrow;
        u[i] = (uint32_t)d;
        borrow = (d >> 63) & 1;
      }
      mul_low(x.data(), u.data(), t.data());
      x = t;
    }
    for (uint32_t i = 0; i < N; ++i) a.pinv[i] = x[i];
  }
  hipStream_t st = (hipStream_t)stream_;
  void *d_calls = nullptr, *d_consts = nullptr, *d_out = nullptr;
  HIP_OK(hipMalloc(&d_calls, calls4.size() * 4));
  HIP_OK(hipMalloc(&d_consts, std::max<size_t>(const_words.size() * 4, 64)));
  HIP_OK(hipMalloc(&d_out, out->size()));
  HIP_OK(hipMemcpy(d_calls, calls4.data(), calls4.size() * 4, hipMemcpyHostToDevice));
  if (!const_words.empty()) HIP_OK(hipMemcpy(d_consts, const_words.data(), const_words.size() * 4, hipMemcpyHostToDevice));
  a.calls = (const zkgpu::R1csCorrCall*)d_calls;
  a.n_calls = n;
  a.table = (const uint4*)d_table_;
  a.n_slots = table_slots_;
  a.batch = batch_;
  a.consts = (const zkgpu::u32*)d_consts;
  a.out = (zkgpu::u32*)d_out;
  const dim3 grid((n + 3) / 4, lane_blocks_);
  switch (nwords_) {
#define X(W) case W: zkgpu::launch_r1cs_corr_w##W(grid, st, a, fp); break;
    ZK_WIDTHS(X)
#undef X
    default: throw std::runtime_error("Engine: unsupported limb count");
  }
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipMemcpy(out->data(), d_out, out->size(), hipMemcpyDeviceToHost));
  (void)hipFree(d_calls);
  (void)hipFree(d_consts);
  (void)hipFree(d_out);
}

void Engine::r1cs_finish_check() {
  use_device();
  hipStream_t st = (hipStream_t)stream_;
  zkgpu::launch_verdict(dim3((batch_ + 255) / 256), st, (const zkgpu::u32*)d_r1cs_fail_, (const zkgpu::u32*)verdict_flags(), batch_,
                        (unsigned long long*)d_r1cs_counts_);
  HIP_OK(hipEventRecord((hipEvent_t)ev_r1cs_end_, st));
  HIP_OK(hipGetLastError());
}

void Engine::r1cs_results(std::vector<uint32_t>* first_fail_row, uint64_t counts[2]) {
  use_device();
  HIP_OK(hipStreamSynchronize((hipStream_t)stream_));
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, (hipEvent_t)ev_r1cs_begin_, (hipEvent_t)ev_r1cs_end_) == hipSuccess) last_r1cs_ms_ = ms;
  if (first_fail_row) {
    first_fail_row->resize(batch_);
    HIP_OK(hipMemcpy(first_fail_row->data(), d_r1cs_fail_, (size_t)batch_ * 4, hipMemcpyDeviceToHost));
  }
  if (counts) HIP_OK(hipMemcpy(counts, d_r1cs_counts_, 16, hipMemcpyDeviceToHost));
}

void Engine::synchronize() {
  use_device();
  HIP_OK(hipStreamSynchronize((hipStream_t)stream_));
  if (d_stamps_ && !stamps_path_.empty()) {
    std::vector<unsigned long long> h(zkgpu::kStampLevels * 16);
    HIP_OK(hipMemcpy(h.data(), d_stamps_, h.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* f = fopen(stamps_path_.c_str(), "wb")) {
      fwrite(h.data(), 8, h.size(), f);
      fclose(f);
    }
  }
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, (hipEvent_t)ev_begin_, (hipEvent_t)ev_end_) == hipSuccess) last_ms_ = ms;
}

void Engine::download(std::vector<uint32_t>* first_fail, std::vector<uint32_t>* flags, uint64_t counts[2]) {
  synchronize();   // (makes the engine's device current)
  if (first_fail) {
    first_fail->resize(batch_);
    HIP_OK(hipMemcpy(first_fail->data(), verdict_first_fail(), (size_t)batch_ * 4, hipMemcpyDeviceToHost));
  }
  if (flags) {
    flags->resize(batch_);
    HIP_OK(hipMemcpy(flags->data(), verdict_flags(), (size_t)batch_ * 4, hipMemcpyDeviceToHost));
  }
  if (counts) HIP_OK(hipMemcpy(counts, verdict_counts(), 16, hipMemcpyDeviceToHost));
}

void Engine::read_input(uint32_t stream, uint32_t position, std::vector<uint8_t>* out, uint32_t* width) {
  use_device();
  if (!batch_) throw std::runtime_error("Engine: no batch");
  const uint32_t n_vals = stream == 0 ? n_inst_ : stream == 1 ? n_wit_ : n_carry_;
  const uint32_t w = stream == 2 ? 4 * carry_words_ : in_stride_;
  const uint8_t* base = (const uint8_t*)(stream == 0 ? d_inst_ : stream == 1 ? d_wit_ : d_carry_);
  if (stream > 2 || position >= n_vals || !base) throw std::runtime_error("Engine: no such input value");
  synchronize();
  if (copy_stream_) HIP_OK(hipStreamSynchronize((hipStream_t)copy_stream_));   // (an upload still in flight)
  out->assign((size_t)batch_ * w, 0);
  HIP_OK(hipMemcpy2D(out->data(), w, base + (size_t)position * w, (size_t)n_vals * w, w, batch_, hipMemcpyDeviceToHost));
  *width = w;
}

void Engine::dump_slots(const std::vector<uint32_t>& slots, std::vector<uint8_t>* out) {
  use_device();
  synchronize();
  const uint32_t k = (uint32_t)slots.size();
  out->assign((size_t)batch_ * k * elem_bytes_, 0);
  if (!k) return;
  hipStream_t st = (hipStream_t)stream_;
  uint32_t* d_slots = nullptr;
  void* d_out = nullptr;
  HIP_OK(hipMalloc(&d_slots, (size_t)k * 4));
  HIP_OK(hipMalloc(&d_out, out->size()));
  HIP_OK(hipMemcpy(d_slots, slots.data(), (size_t)k * 4, hipMemcpyHostToDevice));
  zkgpu::FieldParams fp;
  memcpy(&fp, field_params_, sizeof fp);
  const uint32_t lb64 = (batch_ + 63) / 64;
  // grid.x is limited to 2^31-1, grid.y to 65535: chunk the slot list
  if (boolean_) {
    zkgpu::launch_bool_dump(dim3(k, lb64), st, (const zkgpu::u64*)d_table_, sched_.n_slots, d_slots, k, batch_, (uint8_t*)d_out);
  } else if (generic_) {
    zkgpu::launch_dump_generic(dim3(k, lb64), st, (const uint4*)d_table_, table_slots_, d_slots, k, batch_, (zkgpu::u32*)d_out, nwords_);
  } else {
    launch_dump(nwords_, dim3(k, lb64), st, (const uint4*)d_table_, table_slots_, d_slots, k, batch_, (zkgpu::u32*)d_out, fp);
  }
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipMemcpy(out->data(), d_out, out->size(), hipMemcpyDeviceToHost));
  (void)hipFree(d_slots);
  (void)hipFree(d_out);
}

// ---- field segments of one session (capi.cpp): engines chained on ONE stream, sharing the verdict words of the first ----
void Engine::chain_to(Engine* head, bool first, bool last) {
  chain_first_ = first;
  chain_last_ = last;
  graph_dirty_ = true;
  if (head == this || head == nullptr) {
    chain_head_ = nullptr;
    return;
  }
  if (head->device_ != device_) throw std::runtime_error("Engine: chained engines must live on one device");
  chain_head_ = head;
  if (stream_ != head->stream_) {   // everything this engine enqueues is ordered behind the segment before it
    if (!owned_stream_) owned_stream_ = stream_;
    stream_ = head->stream_;
  }
}

void* Engine::verdict_first_fail() const { return chain_head_ ? chain_head_->d_first_fail_ : d_first_fail_; }
void* Engine::verdict_flags() const { return chain_head_ ? chain_head_->d_flags_ : d_flags_; }
void* Engine::verdict_counts() const { return chain_head_ ? chain_head_->d_counts_ : d_counts_; }

void Engine::set_input_stride(uint32_t bytes) {
  if (batch_) throw std::runtime_error("Engine: set_input_stride() must precede set_batch()");
  in_stride_ = bytes;
  in_stride_set_ = true;
}

// The canonical values of `slots` for every lane, written by THIS engine (on the shared stream, behind its replay) into
// the carry buffer of the engine that runs the next field segment: [lane][k][nwords of this field].
void Engine::carry_out(const std::vector<uint32_t>& slots, Engine* next) {
  use_device();
  if (boolean_ || next->boolean_) throw std::runtime_error("Engine: values carried between field segments need arithmetic fields");
  if (next->n_carry_ != slots.size() || next->batch_ != batch_)
    throw std::runtime_error("Engine: the next segment does not expect these carried values");
  if (slots.empty()) return;   // (everything that crossed the field change was an input or a constant: capi.cpp switch_field)
  if (next->carry_words_ != nwords_ || !next->d_carry_)
    throw std::runtime_error("Engine: the next segment does not expect these carried values");
  for (uint32_t sl : slots)
    if (sl >= table_slots_) throw std::runtime_error("Engine: a carried value names a wire-table slot out of range");
  hipStream_t st = (hipStream_t)stream_;
  if (!d_carry_slots_) {
    HIP_OK(hipMalloc(&d_carry_slots_, slots.size() * 4));
    HIP_OK(hipMemcpy(d_carry_slots_, slots.data(), slots.size() * 4, hipMemcpyHostToDevice));
  }
  zkgpu::FieldParams fp;
  memcpy(&fp, field_params_, sizeof fp);
  if (generic_)
    zkgpu::launch_dump_generic(dim3((uint32_t)slots.size(), (batch_ + 63) / 64), st, (const uint4*)d_table_, table_slots_,
                               (const uint32_t*)d_carry_slots_, (uint32_t)slots.size(), batch_, (zkgpu::u32*)next->d_carry_, nwords_);
  else
    launch_dump(nwords_, dim3((uint32_t)slots.size(), (batch_ + 63) / 64), st, (const uint4*)d_table_, table_slots_,
                (const uint32_t*)d_carry_slots_, (uint32_t)slots.size(), batch_, (zkgpu::u32*)next->d_carry_, fp);
  HIP_OK(hipGetLastError());
}

int current_device() {
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return d;
}

int visible_devices() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return n;
}

// ---- RCCL count reduction for one process driving several GPUs -------------------------------------
namespace {
typedef int (*nccl_comm_init_all_t)(ncclComm_t*, int, const int*);
typedef int (*nccl_all_reduce_t)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
typedef int (*nccl_group_t)(void);
typedef int (*nccl_comm_destroy_t)(ncclComm_t);
typedef const char* (*nccl_error_string_t)(int);
}  // namespace

CountReducer::CountReducer(const std::vector<Engine*>& engines) : engines_(engines) {
  std::vector<int> devs;
  for (Engine* e : engines_) {
    if (e->device() < 0) throw std::runtime_error("CountReducer: engine without a device of its own");
    for (int d : devs)
      if (d == e->device()) throw std::runtime_error("CountReducer: RCCL needs distinct devices");
    devs.push_back(e->device());
  }
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    lib_ = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (lib_) break;
  }
  if (!lib_) throw std::runtime_error(std::string("RCCL: cannot load librccl.so: ") + dlerror());
  const char* names[6] = {"ncclCommInitAll", "ncclAllReduce", "ncclGroupStart", "ncclGroupEnd", "ncclCommDestroy", "ncclGetErrorString"};
  for (int k = 0; k < 6; ++k) {
    fn_[k] = dlsym(lib_, names[k]);
    if (!fn_[k]) throw std::runtime_error(std::string("RCCL: symbol missing: ") + names[k]);
  }
  std::vector<ncclComm_t> comms(devs.size());
  const int rc = ((nccl_comm_init_all_t)fn_[0])(comms.data(), (int)devs.size(), devs.data());
  if (rc != 0) throw std::runtime_error(std::string("RCCL: ncclCommInitAll: ") + ((nccl_error_string_t)fn_[5])(rc));
  for (ncclComm_t c : comms) comms_.push_back((void*)c);
  for (Engine* e : engines_) {
    HIP_OK(hipSetDevice(e->device()));
    void* d = nullptr;
    HIP_OK(hipMalloc(&d, 16));
    reduced_.push_back(d);
  }
}

CountReducer::~CountReducer() {
  for (size_t k = 0; k < reduced_.size(); ++k) {
    (void)hipSetDevice(engines_[k]->device());
    (void)hipFree(reduced_[k]);
  }
  if (fn_[4])
    for (void* c : comms_) (void)((nccl_comm_destroy_t)fn_[4])((ncclComm_t)c);
  // the library stays loaded: RCCL keeps threads of its own
}

void CountReducer::all_reduce(uint64_t totals[2]) {
  auto check_nccl = [&](int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string("RCCL: ") + what + ": " + ((nccl_error_string_t)fn_[5])(rc));
  };
  // whatever happens between ncclGroupStart and ncclGroupEnd, the group is closed again: an open group would swallow
  // every later collective of this thread
  struct GroupGuard {
    nccl_group_t end;
    bool open = false;
    ~GroupGuard() {
      if (open) (void)end();
    }
  } group{(nccl_group_t)fn_[3]};
  check_nccl(((nccl_group_t)fn_[2])(), "ncclGroupStart");
  group.open = true;
  for (size_t k = 0; k < engines_.size(); ++k) {
    HIP_OK(hipSetDevice(engines_[k]->device()));
    // on the engine's stream: ordered behind the verdict kernel of its replay
    check_nccl(((nccl_all_reduce_t)fn_[1])(engines_[k]->counts_device(), reduced_[k], 2, ncclUint64, ncclSum, (ncclComm_t)comms_[k],
                                            (hipStream_t)engines_[k]->stream()),
               "ncclAllReduce");
  }
  group.open = false;
  check_nccl(((nccl_group_t)fn_[3])(), "ncclGroupEnd");
  for (Engine* e : engines_) e->synchronize();
  HIP_OK(hipSetDevice(engines_[0]->device()));
  HIP_OK(hipMemcpy(totals, reduced_[0], 16, hipMemcpyDeviceToHost));
  ++n_reductions_;
}

}  // namespace zki
