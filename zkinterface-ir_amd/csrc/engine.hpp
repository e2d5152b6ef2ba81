// Device engine: owns the HBM-resident program (scheduled tape + constant
// pool), the witness-major wire table and the per-witness verdict words, and
// replays the program for a batch on one MI355X.  No host fallback: every
// entry point throws if the HIP runtime or a GPU is missing.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>

#include "r1cs.hpp"
#include "schedule.hpp"

namespace zki {

struct LaunchTiming {
  uint32_t launch = 0;  // index into Schedule::launches
  uint32_t count = 0;   // ops in the launch
  float ms = 0.f;
};

class Engine {
 public:
  // device < 0: whatever device is current on the calling thread (the usual single-GPU process); otherwise the
  // engine lives on that device and every entry point makes it current on the calling thread first, so that one
  // process can drive several GPUs, each from a host thread of its own (capi.cpp, option "devices")
  explicit Engine(int device = -1);
  ~Engine();
  int device() const { return device_; }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  // Streaming ingest: the program entries of one tape window, sent to HBM while later windows are still being
  // parsed and scheduled.  `entries` = DevOp2 (fused format) or DevOp records; windows arrive in order.  A following
  // load_program() of the finished schedule keeps what was sent (it must be the same windows) and sends the rest.
  void upload_window(const void* entries, uint64_t n_entries, size_t entry_bytes);
  size_t windows_uploaded() const { return d_windows_.size(); }
  // Upload the program.  n_instance / n_witness = values per witness stream.
  static void validate_program(const Schedule& s, uint32_t n_instance, uint32_t n_witness, uint32_t n_carry = 0);
  // n_carry values of carry_words 32-bit words each arrive from the previous field segment of the session (TK_CARRY)
  void load_program(const Schedule& s, const FieldHost& f, uint32_t n_instance, uint32_t n_witness, uint32_t n_carry = 0,
                    uint32_t carry_words = 0);
  // The arithmetic of the any-modulus kernels (device/generic_kernels.hpp) run on the HOST -- the same functions -- for
  // the CPU-tier tests: op 0 add, 1 mul, 2 reduce(a), 3 and, 4 xor over f.nwords words each.  Returns non-zero if `f`
  // is not a generic field or the op is unknown.
  static int generic_selftest(const FieldHost& f, int op, const uint32_t* a, const uint32_t* b, uint32_t* out);
  // Bytes per input value the batch buffers must use: 4*nwords (arithmetic) or 1 (GF(2)).
  uint32_t elem_bytes() const { return elem_bytes_; }

  // (Re)size the per-batch state.  Inputs are [batch][n][elem_bytes] little-endian.
  void set_batch(uint32_t batch);
  void upload_inputs(const uint8_t* inst, const uint8_t* wit);        // host -> HBM
  void use_device_inputs(const void* d_inst, const void* d_wit);      // already resident
  // Limit how many lanes are replayed together (0 = all): lane groups run one
  // after the other so that a group's live wires stay in the 256 MiB Infinity Cache.
  void set_lane_group(uint32_t lanes) { lane_group_ = lanes; graph_dirty_ = true; }
  void set_graph_mode(int mode) { graph_mode_ = mode; graph_dirty_ = true; }  // 0 off (default), 1 on
  // Replay the lane blocks as `n` interleaved halves on `n` HIP streams (1..4): the levels of one
  // half fill the kernel-boundary bubbles and wave tails of the other.
  void set_xcd_map(bool on) { xcd_map_ = on; graph_dirty_ = true; }
  void set_level_ops_per_wave(uint32_t n) { level_ops_per_wave_ = n < 1 ? 1 : (n > 8 ? 8 : n); graph_dirty_ = true; }
  void set_hot_waves(uint32_t n) { hot_waves_ = n; graph_dirty_ = true; }  // 0 = whatever the registers allow
  void set_streams(uint32_t n) { n_streams_ = n < 1 ? 1 : (n > kMaxStreams ? kMaxStreams : n); graph_dirty_ = true; }
  static constexpr uint32_t kMaxStreams = 4;
  static constexpr uint64_t kInfinityCacheBudget = 288ull << 20;  // wire-table bytes kept in flight per lane group
  // GF(2): 0 = pick automatically, 1 = force the HBM-table kernel, 2 = require the LDS-resident kernel
  void set_bool_path(int mode) { bool_path_ = mode; }
  bool uses_lds_path() const { return lds_path_; }

  void replay(bool time_each_launch = false);  // asynchronous on the engine's stream
  void synchronize();
  float last_replay_ms() const { return last_ms_; }
  const std::vector<LaunchTiming>& launch_timings() const { return timings_; }

  void download(std::vector<uint32_t>* first_fail, std::vector<uint32_t>* flags, uint64_t counts[2]);
  void* counts_device() const { return verdict_counts(); }  // u64[2] {satisfied, failed} (of the whole chain of field segments)
  void* stream() const { return stream_; }
  // out[lane][k][elem_bytes]: canonical value of slot slots[k] for every lane
  void dump_slots(const std::vector<uint32_t>& slots, std::vector<uint8_t>* out);
  // out[lane][*width bytes]: the RAW value every lane handed over at `position` of input stream 0 (instance), 1 (witness)
  // or 2 (carried in from the previous field segment): what Evaluator::get returns for a wire that is a copy of an input
  void read_input(uint32_t stream, uint32_t position, std::vector<uint8_t>* out, uint32_t* width);

  // ---- R1CS rows over the same wire table (arithmetic fields) ----------------------------------
  // extra table slots behind the program's own (variables assigned by r1cs_run(assign=true)); call
  // before set_batch()
  void reserve_extra_slots(uint32_t n);
  uint32_t program_slots() const { return sched_.n_slots; }
  void r1cs_upload(const std::vector<R1csRowDev>& rows, const std::vector<R1csTermDev>& terms,
                   const std::vector<uint32_t>& coef_words);
  void r1cs_begin_check();                                        // reset the failing-row words
  void r1cs_run(bool assign, uint32_t first_row, uint32_t n_rows);  // asynchronous
  void r1cs_finish_check();                                       // counts; records the event time
  void r1cs_results(std::vector<uint32_t>* first_fail_row, uint64_t counts[2]);
  float last_r1cs_ms() const { return last_r1cs_ms_; }
  // quotient ("correction") wires of the R1CS conversion for the listed calls: calls4 = {slot a, slot b | constant
  // index, slot out, flags (1 mul, 2 b is a constant)} per call, const_words = the raw constants (nwords each);
  // out[lane][call][elem_bytes] little-endian.  Needs the retain_all wire table of a replay.
  void r1cs_corrections(const std::vector<uint32_t>& calls4, const std::vector<uint32_t>& const_words, std::vector<uint8_t>* out);

  uint64_t table_bytes() const { return table_bytes_; }
  uint32_t batch() const { return batch_; }

  // ---- field segments of one session: a relation whose modulus changes between messages (evaluator.rs:232-237) is a chain
  // of engines, one per field, replayed one after the other on the stream of the first (`head`), all writing the
  // verdict words of the first; the wires alive at a boundary travel as canonical integers (carry_out -> TK_CARRY).
  void chain_to(Engine* head, bool first, bool last);
  // bytes per input value in the caller's buffers when that is wider than this field's limbs (the widest field of the
  // session); before set_batch()
  void set_input_stride(uint32_t bytes);
  uint32_t input_stride() const { return in_stride_; }
  const void* device_instances() const { return d_inst_; }
  const void* device_witnesses() const { return d_wit_; }
  void carry_out(const std::vector<uint32_t>& slots, Engine* next);

 private:
  int device_ = -1;
  void use_device() const;   // make device_ current on this thread
  void free_batch();
  void launch_range(uint32_t lb0, uint32_t lbs, bool time_each);
  void launch_one(size_t li, uint32_t lb0, uint32_t lbs, void* stream);
  void staged_upload(void* dst, const uint8_t* src, size_t bytes);

  Schedule sched_;  // host copy (launch list)
  void* d_stamps_ = nullptr;        // developer instrumentation (ZKGPU_STRAND_STAMPS)
  std::string stamps_path_;
  bool loaded_ = false;
  bool boolean_ = false;
  bool generic_ = false;           // canonical residues, the any-modulus kernels (FieldHost::generic)
  void* d_generic_params_ = nullptr;   // zkgpu::GenericParams
  uint32_t generic_k_words_ = 0;       // words of the characteristic (which instantiation of the any-modulus kernel runs)
  uint32_t nwords_ = 0, elem_bytes_ = 0;
  uint32_t n_inst_ = 0, n_wit_ = 0;
  uint32_t batch_ = 0, lane_blocks_ = 0, lanes_per_block_ = 64, lane_group_ = 0;
  uint64_t table_bytes_ = 0;
  float last_ms_ = 0.f;
  std::vector<LaunchTiming> timings_;

  void* stream_ = nullptr;
  void* side_streams_[3] = {nullptr, nullptr, nullptr};
  void* ev_fork_ = nullptr;
  void* ev_join_[3] = {nullptr, nullptr, nullptr};
  uint32_t n_streams_ = 2;
  bool xcd_map_ = true;
  int graph_mode_ = 0;
  bool graph_dirty_ = true;
  void* graph_exec_ = nullptr;
  bool use_graph() const;
  void capture_graph();
  void enqueue_replay(bool time_each_launch);
  uint32_t level_ops_per_wave_ = 1;
  uint32_t hot_waves_ = 0;
  void* ev_begin_ = nullptr;
  void* ev_end_ = nullptr;
  std::vector<void*> launch_events_;
  std::vector<void*> d_windows_;            // program entries, one device buffer per tape window
  std::vector<uint64_t> window_entries_;    // entries in each
  size_t window_entry_bytes_ = 0;
  void free_windows();
  void* d_consts_ = nullptr;
  void* d_level_ptr_ = nullptr;     // level bounds of the strands (Schedule::strand_level_ptr)
  void* d_table_ = nullptr;
  void* d_first_fail_ = nullptr;
  void* d_flags_ = nullptr;
  void* d_counts_ = nullptr;
  // Engine-owned input buffers, two sets: an upload fills the set the replay in flight is not reading, on its own
  // copy stream, so handing over batch k+1 overlaps the replay of batch k.
  void* d_inst_own_[2] = {nullptr, nullptr};
  void* d_wit_own_[2] = {nullptr, nullptr};
  int own_set_ = 0;                          // set the last upload went to
  void* copy_stream_ = nullptr;
  void* ev_upload_ = nullptr;                // end of the last upload (the next replay waits for it)
  void* ev_set_free_[2] = {nullptr, nullptr};  // end of the last replay that read set k
  bool upload_pending_ = false;
  bool set_read_[2] = {false, false};
  const void* d_inst_ = nullptr;
  const void* d_wit_ = nullptr;
  void* d_strict_inst_ = nullptr;  // per input position, the mode of a value >= p (Schedule::strict_instance)
  void* d_strict_wit_ = nullptr;
  void* d_strict_carry_ = nullptr;
  void* d_carry_ = nullptr;        // [lane][n_carry][carry_words]: values carried in from the previous field segment
  void* d_carry_slots_ = nullptr;  // slots this engine carries out (device copy)
  void* d_input_aux_ = nullptr;    // zkgpu::InputAux of this batch
  uint32_t n_carry_ = 0, carry_words_ = 0;
  uint32_t in_stride_ = 0;
  bool in_stride_set_ = false;
  Engine* chain_head_ = nullptr;
  bool chain_first_ = true, chain_last_ = true;
  void* owned_stream_ = nullptr;   // this engine's own stream while it runs on the chain head's
  void* verdict_first_fail() const;
  void* verdict_flags() const;
  void* verdict_counts() const;
  void* d_packed_inst_ = nullptr;  // GF(2) path
  void* d_packed_wit_ = nullptr;
  void* h_stage_[2] = {nullptr, nullptr};  // pinned staging for host -> HBM input uploads
  void* ev_stage_[2] = {nullptr, nullptr};
  bool stage_used_[2] = {false, false};
  int stage_next_ = 0;
  void* d_r1cs_rows_ = nullptr;
  void* d_r1cs_terms_ = nullptr;
  void* d_r1cs_coefs_ = nullptr;
  void* d_r1cs_fail_ = nullptr;
  void* d_r1cs_counts_ = nullptr;
  void* ev_r1cs_begin_ = nullptr;
  void* ev_r1cs_end_ = nullptr;
  uint32_t r1cs_rows_ = 0, r1cs_one_coef_ = 0;
  bool r1cs_classes_ = false;   // some combination of the rows is of class unit / small (device/args.hpp)
  uint32_t extra_slots_ = 0, table_slots_ = 0;
  float last_r1cs_ms_ = 0.f;
  void* d_lds_ops_ = nullptr;       // 8-byte program entries of the LDS-resident GF(2) kernel (generic chunks)
  void* d_lds_ops6_ = nullptr;      // its rows of xor / and / not / copy ops: 6 bytes per op
  void* d_lds_blocks_ = nullptr;    // block headers of those rows
  uint32_t lds_block_rows_ = 0;     // rows per block (selects the kernel instantiation)
  void* d_launches_ = nullptr;
  uint32_t n_lds_chunks_ = 0;
  int bool_path_ = 0;
  bool lds_path_ = false;
  bool lds_writeback_ = false;
  bool force_writeback_ = false;
 public:
  // pinned wires (Evaluator::get) need the LDS-resident values written back to HBM
  void set_writeback(bool on) { force_writeback_ = on; graph_dirty_ = true; }
 private:
  unsigned char field_params_[256];  // zkgpu::FieldParams, opaque here
};

// One process driving several GPUs: the {satisfied, failed} counters of engines on DISTINCT devices are combined by an
// RCCL all-reduce over xGMI (ncclCommInitAll: one communicator per device, all owned by this process), enqueued on
// every engine's own stream behind its replay; afterwards every device holds the totals.  RCCL is loaded on first
// use (dlopen: the library takes no link-time dependency on it).  Throws when RCCL cannot be loaded or a call fails.
int visible_devices();   // hipGetDeviceCount, or -1 when the HIP runtime finds no GPU
int current_device();    // hipGetDevice of the calling thread, or -1 without a GPU

class CountReducer {
 public:
  explicit CountReducer(const std::vector<Engine*>& engines);
  ~CountReducer();
  CountReducer(const CountReducer&) = delete;
  CountReducer& operator=(const CountReducer&) = delete;
  void all_reduce(uint64_t totals[2]);   // enqueue, wait, read the totals back from the first device
  uint64_t reductions() const { return n_reductions_; }   // all-reduces that ran to the end

 private:
  std::vector<Engine*> engines_;
  std::vector<void*> comms_;      // ncclComm_t per engine
  std::vector<void*> reduced_;    // device u64[2] per engine
  void* lib_ = nullptr;
  void* fn_[6] = {nullptr};
  uint64_t n_reductions_ = 0;
};

}  // namespace zki
