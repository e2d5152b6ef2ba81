/* zkgpu.h -- C ABI of the MI355X-native SIEVE IR batch evaluator.
 *
 * Drop-in boundary for the `evaluate` / `valid-eval-metrics` hot path of
 * dryajov/zkinterface-ir (zki_sieve 3.0.0).  Three groups of entry points:
 *
 *  1. zkgpu_backend_*  -- one function per method of the reference's plugin
 *     trait `ZKBackend` (rust/src/consumers/evaluator.rs:17-76).  A Rust
 *     `impl ZKBackend for GpuBackend` binds them 1:1; the
 *     reference's own `Evaluator` then records the circuit into the GPU tape,
 *     exactly as it drives `IRFlattener` (rust/src/consumers/flattening.rs:42-191).
 *  2. zkgpu_ingest_* / zkgpu_declare_inputs -- the `Evaluator` entry points
 *     (`from_messages`, `ingest_message`; evaluator.rs:187-303) and `Source`
 *     (rust/src/consumers/source.rs:59-118) for callers without a Rust
 *     toolchain: size-prefixed `.sieve` FlatBuffers in, tape out.
 *  3. zkgpu_finalize / zkgpu_set_inputs* / zkgpu_replay / result getters --
 *     the batch extension: the recorded tape is replayed by HIP kernels for
 *     `batch` independent (instance, witness) pairs, one lane per witness.
 *     Lane i behaves like the i-th run of `zki_sieve evaluate` on the same
 *     relation (cli.rs:315-320): same verdict, same violation strings.
 *
 * Conventions: plain pointers and sizes, no ownership transfer; every byte
 * buffer is borrowed for the duration of the call.  Functions returning int
 * return 0 on success and non-zero on failure, with text in zkgpu_last_error().
 * A session is single-owner and not thread-safe (the reference `Evaluator`
 * is `&mut self` everywhere).  There is no CPU fallback: replay entry points
 * fail if no GPU is present.
 */
#ifndef ZKGPU_H
#define ZKGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zkgpu_session zkgpu_session;

#define ZKGPU_NO_FAIL 0xFFFFFFFFu        /* first_fail value of a satisfied lane */
#define ZKGPU_LANE_NONCANONICAL 0x1u     /* lane flag: an input value was >= p   */

/* ---- lifecycle ---------------------------------------------------------- */
zkgpu_session* zkgpu_session_new(void);                 /* Evaluator::default() + backend */
void zkgpu_session_free(zkgpu_session* s);
const char* zkgpu_last_error(const zkgpu_session* s);
const char* zkgpu_version(void);

/* ---- 1. ZKBackend trait (evaluator.rs:17-76) ----------------------------- */
/* Wires are uint32 handles owned by the session (like IRFlattener::Wire = WireId); they are session-wide numbers, never
 * reused.
 * set_field: the reference's Evaluator calls it for every Relation message with whatever modulus the header holds
 * (evaluator.rs:262-268).  The same modulus again only updates is_boolean.  ANOTHER modulus opens a new field segment
 * (see "Field segments" below): the wires that live on are the handles the caller has not reported dropped
 * (zkgpu_backend_drop) -- they keep their numbers and hold, in the new field, the integers they held; a handle that was
 * dropped before the change is an error to name afterwards.  A binding whose Wire type does not implement Drop keeps
 * every value alive, and every value would have to be carried: refused beyond 2^20 live wires, with a text that says so. */
int zkgpu_backend_set_field(zkgpu_session* s, const uint8_t* modulus_le, size_t len, uint32_t degree,
                            int is_boolean);                                   /* :28 */
int zkgpu_backend_copy(zkgpu_session* s, uint32_t wire, uint32_t* out);        /* :38 */
int zkgpu_backend_constant(zkgpu_session* s, const uint8_t* value_le, size_t len, uint32_t* out); /* :41 */
/* local_wire_id = the id printed in "Wire_{} (may be weighted) should be 0, while it is not" (:357-363) */
int zkgpu_backend_assert_zero(zkgpu_session* s, uint32_t wire, uint64_t local_wire_id);          /* :46 */
int zkgpu_backend_add(zkgpu_session* s, uint32_t a, uint32_t b, uint32_t* out);                  /* :49 */
int zkgpu_backend_multiply(zkgpu_session* s, uint32_t a, uint32_t b, uint32_t* out);             /* :51 */
int zkgpu_backend_add_constant(zkgpu_session* s, uint32_t a, const uint8_t* c_le, size_t len, uint32_t* out); /* :53 */
int zkgpu_backend_mul_constant(zkgpu_session* s, uint32_t a, const uint8_t* c_le, size_t len, uint32_t* out); /* :55 */
int zkgpu_backend_and(zkgpu_session* s, uint32_t a, uint32_t b, uint32_t* out);                  /* :58 */
int zkgpu_backend_xor(zkgpu_session* s, uint32_t a, uint32_t b, uint32_t* out);                  /* :60 */
int zkgpu_backend_not(zkgpu_session* s, uint32_t a, uint32_t* out);                              /* :62 */
/* instance()/witness() (:66,:75): the FieldElement is the position of the value in the lane's
 * instance / witness stream; the values themselves arrive per lane through zkgpu_set_inputs. */
int zkgpu_backend_instance(zkgpu_session* s, uint32_t position, uint32_t* out);
int zkgpu_backend_witness(zkgpu_session* s, uint32_t position, uint32_t* out);
/* Optional hint from the caller's Evaluator (rust/src/consumers/evaluator.rs:823-839 compute_weight): the backend calls
 * made since zkgpu_tape_len() read `first_call` computed result = base^(modulus - 1), the Switch indicator.  With the
 * hint the production schedule may evaluate the ladder as one `base != 0` entry when the characteristic is prime
 * (option "fermat"); without it the ladder is replayed product by product.  The bundled Evaluator gives it itself. */
int zkgpu_backend_ladder(zkgpu_session* s, uint64_t first_call, uint32_t base, uint32_t result);
/* Optional: the caller let go of `wire` -- no later call will name it.  This is `impl Drop for` the Rust binding's Wire
 * type: the reference's Evaluator owns its wires (`HashMap<WireId, B::Wire>`, evaluator.rs:158-185) and drops them on
 * `Free`, at sub-circuit scope exit and at the end of expressions.  The streaming scheduler (option "stream") uses the
 * drops to recycle wire-table slots and fuse gates before the rest of the relation has arrived; without them every
 * value stays materialised until zkgpu_finalize.  The bundled Evaluator reports drops itself. */
int zkgpu_backend_drop(zkgpu_session* s, uint32_t wire);

/* ---- 2. Evaluator / Source entry points ---------------------------------- */
/* A byte stream of one or more size-prefixed messages (consumers/utils.rs:6-41).
 * Instance / Witness messages define lane 0's input streams (single-statement use). */
int zkgpu_ingest_messages(zkgpu_session* s, const uint8_t* data, size_t len);
/* Files and directories, read in the order of Source::from_dirs_and_files (source.rs:64-89). */
int zkgpu_ingest_paths(zkgpu_session* s, const char* const* paths, size_t n_paths);
/* Batch use: announce n_instance / n_witness queued values per lane before the Relation
 * messages are ingested (what ingest_instance / ingest_witness do for one statement). */
int zkgpu_declare_inputs(zkgpu_session* s, uint32_t n_instance, uint32_t n_witness);
/* Evaluator::get_violations() of the recording itself (lane independent part). */
size_t zkgpu_host_violations(zkgpu_session* s, char* buf, size_t cap);

/* tape inspection (host logic tests, statistics) */
uint64_t zkgpu_tape_len(const zkgpu_session* s);        /* backend calls, asserts included   */
uint64_t zkgpu_tape_value_ops(const zkgpu_session* s);  /* value-returning calls             */
uint64_t zkgpu_tape_asserts(const zkgpu_session* s);
/* kinds[i] in {1 add,2 mul,3 addc,4 mulc,5 copy,6 constant,7 instance,8 witness,9 assert_zero,
 *              10 and,11 xor,12 not}; a[i], b[i] = operand handles / constant index / position */
int zkgpu_tape_dump(const zkgpu_session* s, uint8_t* kinds, uint32_t* a, uint32_t* b, uint64_t cap);
/* local wire id of the k-th assert_zero (the id the violation message prints) */
int zkgpu_tape_assert_wires(const zkgpu_session* s, uint64_t* local_wire_ids, uint64_t cap);
uint32_t zkgpu_n_constants(const zkgpu_session* s);
size_t zkgpu_constant_bytes(const zkgpu_session* s, uint32_t index, uint8_t* out, size_t cap);

/* ---- 3. batch replay on the GPU ------------------------------------------ */
/* Build the device program: levelise, assign wire-table slots (host work; the GPU is first touched
 * by zkgpu_set_inputs*).
 * retain_all != 0 keeps every wire value readable afterwards (parity dumps). */
int zkgpu_finalize(zkgpu_session* s, int retain_all);
/* Streaming ingest (option "stream"): out[0] = tape windows of the program, out[1] = windows that were scheduled (and,
 * with a GPU, uploaded) by the worker thread while messages were still coming in, out[2] = seconds that thread worked. */
int zkgpu_stream_info(const zkgpu_session* s, double out[3]);
uint32_t zkgpu_elem_bytes(const zkgpu_session* s);   /* bytes per input value: 8*limbs, or 1 for GF(2) */
uint32_t zkgpu_n_instance(const zkgpu_session* s);   /* values per lane the tape consumes */
uint32_t zkgpu_n_witness(const zkgpu_session* s);
/* schedule facts: out[0]=levels out[1]=launches out[2]=slots out[3]=widest level out[4]=sequential launches */
/* ... out[5]=device ops out[6]=constant words out[7]=words per constant */
int zkgpu_schedule_info(const zkgpu_session* s, uint64_t out[8]);
/* the device program itself (host-logic tests interpret it without a GPU): ops4 points at 8 words per op
 * {dst, kind | a_expr << 8 | b_expr << 10 | second << 12, a0, a1, b0, b1, dst2, c0} (expr: 0 = the slot,
 * 1 = add(x0,x1), 2 = mul(x0,x1); second != 0: pair entry, also dst2 = add (1) / mul (2) of operand a and slot c0),
 * launches4 = {first,count,ops_per_wave,sequential} per launch, const_words = constant pool in device form,
 * slot_of[i] = wire-table slot of tape op i (0xFFFFFFFF for asserts).  An operand of and / xor over a field other than
 * GF(2) may be 0x80000000 | (2 + 4 * position + stream) instead of a slot: the raw value of that input (stream 0 instance,
 * 1 witness, 2 carried in), of which the operand is only a copy, or (stream 3) of a constant >= the characteristic, which
 * const_words holds as a plain integer behind the zkgpu_n_constants device-form entries, `position` counting from there.
 * Any pointer may be NULL. */
int zkgpu_schedule_dump(const zkgpu_session* s, uint32_t* ops4, uint32_t* launches4, uint32_t* const_words,
                        uint32_t* slot_of);
/* A sequential launch of the fused program is a STRAND: one workgroup per 64 witnesses walks its levels with a barrier
 * between them, entry i of a level going to wave (i - first entry of the level) % 4, a wave running its entries in order.
 * out (cap words; NULL to ask): the level bounds of launch `launch` as entry offsets relative to its first entry (levels + 1
 * numbers), then one word with the LDS-resident values of the strand (slots 0x40000000 | k, k below it).  Returns the
 * number of words (0: not a strand).  For static checks of the schedule (tests/test_strands.py). */
size_t zkgpu_schedule_strand_levels(const zkgpu_session* s, uint32_t launch, uint32_t* out, size_t cap);

/* GF(2) relations: the program of the LDS-resident kernel (csrc/device/lds_layout.hpp) for this schedule, with blocks
 * of `block_rows` rows (4, 6, 8, 9, 10, 12; 0 = the size the engine would pick).  Host work only (no GPU is touched):
 * host-logic tests interpret it.  sizes[6] = {entries of 8 bytes, u16 words of the row stream, block headers, chunks,
 * block_rows used, words of the LDS table}; call once with the array pointers NULL for the sizes, then with buffers
 * (ops8: 4 u16 per entry {dst, a, b, kind}; rows: u16; blocks: 2 u32 per block; chunks: 4 u32 per chunk).  Returns 2
 * (and an error text) when the relation is not Boolean or does not fit the kernel. */
int zkgpu_lds_program(zkgpu_session* s, uint32_t block_rows, uint64_t sizes[6], uint16_t* ops8, uint16_t* rows,
                      uint32_t* blocks, uint32_t* chunks);

/* inputs: [batch][n_instance][elem_bytes] and [batch][n_witness][elem_bytes], little-endian */
int zkgpu_set_inputs(zkgpu_session* s, const uint8_t* instances, const uint8_t* witnesses, uint32_t batch);
int zkgpu_set_inputs_device(zkgpu_session* s, const void* d_instances, const void* d_witnesses, uint32_t batch);
/* use the values of the ingested Instance / Witness messages as a batch of one */
int zkgpu_set_inputs_from_messages(zkgpu_session* s);
/* replay lane groups of this many witnesses one after the other; 0 (default) = automatic: the largest group whose
 * wire table stays in the 256 MiB Infinity Cache (whole XCD rounds per stream), or the whole batch when it fits */
int zkgpu_set_lane_group(zkgpu_session* s, uint32_t lanes);

/* Values >= the field characteristic.  The reference's PlaintextBackend keeps constants, instance and witness values
 * unreduced (rust/src/consumers/evaluator.rs:862-864,896-898,940-946): arithmetic gates reduce their result, but copy
 * clones the integer, assert_zero / not test it for zero, and / xor over a field other than GF(2) work on its bits and
 * Evaluator::get returns it.  After zkgpu_finalize, zkgpu_input_modes gives per input position (witness = 0: instance
 * stream, 1: witness stream; returns the number of positions, writes at most cap bytes) how a value >= p is treated:
 *   0x00  reduced on load: every use is arithmetic (over GF(2): and / xor, whose low bit only depends on low bits);
 *   0x01  read by assert_zero / not alone, through copies: those see "non-zero", as the reference does (a value >= p
 *         is never the integer 0) -- GF(p): the kernels test the raw input beside the wire; GF(2): packed as v != 0;
 *   0x02  GF(p): both of the above (the sinks still get the reference's answer);
 *   0x03  not GF(2): its bits are read as they are -- by and / xor through copies (the entry reads the raw input instead
 *         of the wire) or by zkgpu_get_wire for a wire alive at the end that is a copy of it (answered from the input
 *         itself): the reference's answer again; only a value wider than the limbs of the field flags its lane;
 *   0xFF  over GF(2) (bit-packed) it feeds both a zero test and a gate -- one bit cannot be `v & 1` and `v != 0` -- or it
 *         was carried over two field changes without passing through a gate: a lane holding a value >= p there is
 *         flagged ZKGPU_LANE_NONCANONICAL and counted as failed, with a violation text that says so.
 * Constants >= p get the reference's answer the same way (kept in the pool as the integers they are where their bits
 * are read); zkgpu_finalize refuses one that is wider than the limbs there, and over GF(2) one that feeds a zero test and
 * a gate.
 * The modes depend on the tape alone, not on options ("stream" windows included). */
size_t zkgpu_input_modes(const zkgpu_session* s, int witness, uint8_t* out, size_t cap);

/* options: "max_tape_ops" = N (default 2^30: loops are unrolled, this bounds a corrupt loop bound),
 * "streams" = 1..4 (lane shares replayed concurrently, default 2),
 * "sort_by_operand" = 0|1|2|3 (order of the ops inside a level: tape order, by first operand, that followed by a
 * depth-first walk over shared operands so that the readers of a wire run back to back, or the walk alone; default 3:
 * the pre-sort buys nothing measurable on C2 and costs a third of the scheduling time),
 * "strand_width" = N (levels with fewer than N entries do not get a launch of their own: consecutive ones form a strand
 * that one workgroup per lane block walks with a barrier between levels -- the dependency chains of a structured
 * relation; default 17, 3 = only the levels round 1 walked with a single wave),
 * "strand_lds" = 0|1 (a strand keeps the values that never leave it in the workgroup's LDS; default 1),
 * "strand_prefetch" = 0|1 (what a strand reads out of the wire table is copied into LDS one to three levels ahead, by
 * entries of their own on waves that idle; default 1), "strand_merge" = 0|1 (levels of a strand that need no barrier
 * between them run as one level, a dependent entry behind its producer on the same wave; default 1),
 * "strand_reassociate" = 0|1 (a fused product (z * x) * y whose z comes out of the level in front while x and y have been
 * there for two levels or more keeps z * (x * y), x * y on a spare wave of an earlier level; default 1),
 * "strand_split_inputs" = 0|1 (an instance / witness value of a strand is fetched and checked a level or more ahead of its
 * conversion; default 1) -- all four need "strand_lds",
 * "bool_narrow_width" = 3..2048 (GF(2), LDS-resident kernel: a level of fewer ops than this runs as packets of 64 entries
 * walked by one wave, without a barrier or padded rows; default 257),
 * "r1cs_coef_classes" = 0|1 (the row kernel's cheap paths for combinations whose coefficients are all 1 / -1 or small
 * signed integers, see zkgpu_r1cs_class_counts; set before the rows are made; default 1),
 * "bank_aware" = 0|1 (GF(2): order the ops of a level and number the wire-table slots so that the 32 lanes one LDS
 * instruction serves read and write 32 different banks -- and / xor operands are swapped where that helps; default 1),
 * "graph" = 0|1 (replay the captured hipGraph of the whole launch sequence instead of issuing it launch by launch;
 * default 0: measured slower on ROCm 7.2),
 * "xcd_map" = 0|1 (launches over a multiple of 8 lane blocks give each XCD its own lane blocks; default 1),
 * "level_ops_per_wave" = 1..8 (program entries of a wide level walked by one wave, interleaved over the 4 waves
 * of a workgroup; default 1: with the Add/Mul kernel at 60 VGPRs = 8 waves per SIMD one entry per wave measured fastest),
 * "devices" = "0,1,...,7" (one process driving several GPUs, SURVEY.md 8b/8e: the lanes of a batch are split into
 * contiguous shares of whole lane blocks, one engine per listed HIP device, each fed from a host thread of its own --
 * a replay is hundreds of kernel launches; the program is replicated, no data-path collective.  zkgpu_counts combines
 * the per-device {satisfied, failed} counters with one RCCL all-reduce (ncclCommInitAll; loaded with dlopen on first
 * use) when the listed devices are distinct, and by a host sum when a device is listed more than once -- the
 * rehearsal of the lane split on a one-GPU box -- or when RCCL cannot be loaded or initialised (zkgpu_rccl_note then
 * says why; the counts are exact either way).  The all-reduce over MORE THAN ONE device has not run on hardware yet:
 * no multi-GPU node was available to any round so far; what has run is the one-rank communicator of "force_rccl".  Per-lane results, violations and wire dumps are gathered in lane
 * order.  Not available with several devices: zkgpu_set_inputs_device, zkgpu_counts_device, zkgpu_stream,
 * zkgpu_replay_timed and the R1CS entry points.  Set before the first zkgpu_set_inputs* call.  Default: one engine on
 * the thread's current device),
 * "inspect_segment" = k | "" (which field segment zkgpu_tape_dump, zkgpu_schedule_info / _dump, the constant pool,
 * zkgpu_modulus and zkgpu_input_modes(.., 2, ..) describe; default: the last one),
 * "force_rccl" = 0|1 (zkgpu_counts always goes through ncclCommInitAll + ncclAllReduce, also for a single engine -- a
 * communicator of one rank -- and an RCCL failure is an error instead of falling back to the host sum; default 0),
 * "stream" = 0 | 1 | N (streaming ingest, rust/src/consumers/evaluator.rs:286-301: the reference consumes a relation
 * as a stream of <= 100k-gate messages; with N > 0 the tape is cut into windows of about N recorded calls ("1" = 131072),
 * and a worker thread schedules each window -- and, for the arithmetic kernels, sends its program entries to the GPU --
 * as soon as it is complete, while the caller is still ingesting the following messages; zkgpu_finalize then only
 * schedules the tail.  A window ends in front of the first recorded call that opens a new dependency level once N calls
 * are recorded (a relation recorded level by level is never cut inside a level: its streamed program has the launches of
 * the one scheduled at finalize, and over GF(2) it is that program byte for byte), at 2 N calls at the latest, never
 * inside the exponent ladder of a Switch weight.  The cuts depend on the tape alone, so the program is the same however
 * the relation was split into messages.  What a window may fuse or recycle rests on the drop records of the wires (the
 * bundled Evaluator gives them; zkgpu_backend_drop).  Set before the first Relation message; ignored with retain_all.
 * Default: a relation over GF(2) is streamed (its program is the same either way), any other field is scheduled at
 * finalize -- a streamed schedule of an arithmetic relation replays 1-36 % slower (windows limit fusion and strands),
 * which a single statement does not notice and a batch session does: the one-statement entry points set "1"),
 * "schedule_threads" = N (threads ordering the levels of a window, default min(8, hardware threads)),
 * "hot_waves" = 0 | 3..7 (cap on the resident waves per SIMD of the Add/Mul kernel, by an unused LDS allocation;
 * 0 = no cap, the default -- a tuning handle, every cap measured slower on C2),
 * "fuse" = 0|1 (single-reader Add/Mul gates evaluated inside their reader; never with retain_all),
 * "fermat" = 0|1 (the exponent ladder x^(p-1) of a Switch weight becomes one `x != 0` entry when the characteristic
 * passes the primality test; never with retain_all; default 1),
 * "pair" = 0|1 (an Add/Mul read by exactly two Add/Mul gates of one level is evaluated once inside a pair entry
 * that produces both readers' values; part of "fuse", never with retain_all),
 * "propagate_copies" = 0|1 (readers use a copy's source, unobserved copies are not materialised; never with retain_all),
 * "bool_path" = "auto" | "hbm" | "lds"  (GF(2): HBM wire table, or the whole wire table of a
 * 32-witness slice resident in one CU's LDS when the live wires fit in 160 KiB).  Set before zkgpu_set_inputs*.
 * "validate" = "prover" | "verifier" | "off" and "metrics" = 0|1 switch on the other two consumers of
 * `valid-eval-metrics`; set them before the first zkgpu_ingest_* call ("validator_max_steps" = N bounds the
 * wire-by-wire walk of Free/For ranges, default 2^40). */
int zkgpu_set_option(zkgpu_session* s, const char* key, const char* value);

/* ---- Validator and Stats beside the Evaluator: `zki_sieve valid-eval-metrics` (rust/src/cli.rs:333-363) ----
 * With "validate" / "metrics" enabled every message given to zkgpu_ingest_* is also fed to a Validator
 * (rust/src/consumers/validator.rs:64-861) and a Stats (rust/src/consumers/stats.rs:44-287), in the
 * order the reference feeds them; a message that does not parse makes zkgpu_ingest_* fail (cli.rs:345-346).
 * zkgpu_validator_violations = Validator::get_violations() joined with '\n' (same strings, same order);
 * zkgpu_validator_live_wires = 1 when the reference would print "WARNING: few variables were not freed.";
 * zkgpu_stats_json = serde_json::to_writer_pretty(&stats); zkgpu_stats_warnings = its stderr lines.
 * String getters return the full length and write at most cap-1 bytes + NUL. */
size_t zkgpu_validator_violations(zkgpu_session* s, char* buf, size_t cap);
/* values of the ingested Instance (witness = 0) / Witness (witness = 1) messages, in stream order: what
 * `Evaluator::ingest_instance/ingest_witness` queued (rust/src/consumers/evaluator.rs:239-257).  _values = how
 * many; _value copies up to cap bytes of value `index` (little-endian, as sent) and returns its length. */
uint32_t zkgpu_message_values(const zkgpu_session* s, int witness);
/* the field characteristic the backend was given (`set_field`), little-endian as sent; 0 = no field yet */
size_t zkgpu_modulus(const zkgpu_session* s, uint8_t* buf, size_t cap);
size_t zkgpu_message_value(const zkgpu_session* s, int witness, uint32_t index, uint8_t* buf, size_t cap);
int zkgpu_validator_count(zkgpu_session* s);        /* -1 when the validator is off */
int zkgpu_validator_live_wires(zkgpu_session* s);
size_t zkgpu_stats_json(zkgpu_session* s, char* buf, size_t cap);
size_t zkgpu_stats_warnings(zkgpu_session* s, char* buf, size_t cap);
int zkgpu_uses_lds_path(zkgpu_session* s);           /* 1 / 0, -1 on error (touches the GPU) */
int zkgpu_replay(zkgpu_session* s);                  /* asynchronous */
int zkgpu_replay_timed(zkgpu_session* s);            /* per-launch HIP events, synchronous */
int zkgpu_synchronize(zkgpu_session* s);
float zkgpu_last_replay_ms(const zkgpu_session* s);  /* HIP-event time of the last replay */
/* per-launch timings of the last zkgpu_replay_timed: ms[i], ops[i] for launch i; returns count */
size_t zkgpu_launch_timings(const zkgpu_session* s, float* ms, uint32_t* ops, size_t cap);

/* results (synchronise first) */
int zkgpu_counts(zkgpu_session* s, uint64_t out[2]);          /* {satisfied, failed} */
void* zkgpu_counts_device(zkgpu_session* s);                  /* device uint64[2], for an RCCL all-reduce */
void* zkgpu_stream(zkgpu_session* s);                         /* hipStream_t the replay runs on */
int zkgpu_device_count(void);                                 /* GPUs the HIP runtime sees (-1: none / no runtime) */
int zkgpu_n_engines(const zkgpu_session* s);                  /* engines the batch is split over (option "devices") */
/* Field segments.  The reference takes the modulus afresh from every message header
 * (rust/src/consumers/evaluator.rs:232-237, :262-268): a Relation message may continue under another field characteristic,
 * the wires of the scope live on as the integers they are.  zkgpu_ingest_* open a new FIELD SEGMENT for that -- a
 * backend, a schedule and an engine per field; the wires alive at the boundary become inputs of the next segment
 * (canonical integers written out by the segment before it, subject to the same rules as any unreduced input, see
 * zkgpu_input_modes); the segments replay one after the other on one stream and share the verdict words, assert
 * sequence numbers run through them.  zkgpu_elem_bytes is then the width of the WIDEST field: instance / witness values
 * are handed over in that width whichever segment consumes them (a value that does not fit the limbs of the field that
 * consumes it flags its lane).  A session that changes between GF(2) and another field keeps its GF(2) wires as
 * integers too (the any-modulus kernels, zkgpu_field_representation 2): one value per input position in the common
 * width instead of one byte.  A caller-driven backend changes its field through zkgpu_backend_set_field (the wires that
 * live on: the handles not dropped).  Not supported, with an error that says so: R1CS entry points, zkgpu_replay_timed;
 * option "stream" is switched off at the first change.
 * _info: out = {values carried in, first assert sequence number, 32-bit words per value, values carried out}. */
int zkgpu_n_field_segments(const zkgpu_session* s);
int zkgpu_field_segment_info(const zkgpu_session* s, uint32_t k, uint32_t out[4]);
int zkgpu_field_segment_carried(const zkgpu_session* s, uint32_t k, uint32_t* slots, uint32_t cap);
/* How field segment k keeps a wire on the device: 0 = one bit per witness (GF(2)), 1 = Montgomery form (an odd
 * characteristic of at most 512 bits), 2 = the canonical residue (the any-modulus kernels: an even characteristic, one
 * of up to 4096 bits, GF(2) beside another field; PlaintextBackend takes any BigUint modulus, evaluator.rs:866-938).
 * -1: no such segment, or its field is not set yet. */
int zkgpu_field_representation(const zkgpu_session* s, uint32_t k);
/* Test hook: the modular arithmetic of the any-modulus kernels run on the host (the kernels call the same functions).
 * op 0: out = a + b mod p, 1: a * b mod p (a, b < p), 2: a mod p (a: any value of *nwords words), 3: a & b, 4: (a ^ b) mod p;
 * operands and result hold *nwords 32-bit words (little-endian), which the call also reports (a == NULL: only that).
 * Returns 0, 1 for an unknown op, 2 for a modulus the path does not take (0, 1, more than 4096 bits). */
int zkgpu_generic_selftest(const uint8_t* modulus_le, size_t modulus_len, int op, const uint32_t* a, const uint32_t* b,
                           uint32_t* out, uint32_t* nwords);
uint64_t zkgpu_rccl_reductions(const zkgpu_session* s);       /* zkgpu_counts calls answered by an RCCL all-reduce */
size_t zkgpu_rccl_note(const zkgpu_session* s, char* buf, size_t cap); /* why RCCL was not used ("" = it was, or was not needed) */
int zkgpu_lane_results(zkgpu_session* s, uint32_t* first_fail, uint32_t* flags); /* [batch] each */
/* Evaluator::get_violations() of lane `lane`, '\n'-separated; returns the length needed */
size_t zkgpu_lane_violations(zkgpu_session* s, uint32_t lane, char* buf, size_t cap);
/* retain_all only: out[lane][k][elem_bytes] = value of the k-th value-returning backend call
 * (k in [first, first+count)), i.e. flattened wire k of IRFlattener's numbering. */
int zkgpu_dump_trace_values(zkgpu_session* s, uint64_t first, uint64_t count, uint8_t* out);
/* Evaluator::get(id) (evaluator.rs:750-752) for every lane: out[lane][elem_bytes]; 3 = not found */
int zkgpu_get_wire(zkgpu_session* s, uint64_t wire_id, uint8_t* out);
uint64_t zkgpu_table_bytes(const zkgpu_session* s);

/* ---- 4. R1CS (ir-to-zkif path) ---------------------------------------------------------------------
 * The reference's `ToR1CSConverter` (rust/src/consumers/to_r1cs.rs:93-393) is a ZKBackend that emits one
 * zkinterface BilinearConstraint A*B=C per backend call and, with use_witness, the assignment of every
 * variable.  zkgpu_r1cs_from_tape derives the same constraint system from the recorded tape (variable 0
 * is the constant one, :117; variables are numbered in allocation order); the assignment of a batch is
 * what the replay leaves in the wire table (zkgpu_finalize(retain_all=1)); zkgpu_r1cs_check evaluates
 * <a,w>*<b,w> = <c,w> for every row and lane on the GPU (the job of the zkinterface Simulator the
 * reference's tests call, :583-589). */
int zkgpu_r1cs_from_tape(zkgpu_session* s, int use_correction);
/* out[0]=rows out[1]=variables out[2]=terms out[3]=distinct coefficients */
int zkgpu_r1cs_info(const zkgpu_session* s, uint64_t out[4]);
/* How many of the rows' 3 x rows combinations the row kernel takes by which path: out[0] any coefficients (Montgomery
 * products with the pool), out[1] every coefficient 1 or -1 (additions), out[2] every coefficient a signed integer
 * below 2^31 in magnitude (N word products per term instead of N^2; what FromR1CSConverter expansions mostly hold,
 * from_r1cs.rs:110-125).  Option "r1cs_coef_classes" = "0" (before the rows are made) sends every combination down
 * the first path. */
int zkgpu_r1cs_class_counts(const zkgpu_session* s, uint64_t out[3]);
/* row_ptr: 3 entries per row (start of A, B, C in the term arrays) + final end; var_of_op[i] = variable of
 * tape op i (0xFFFF...F for assert_zero).  Any pointer may be NULL. */
int zkgpu_r1cs_export(const zkgpu_session* s, uint32_t* row_ptr, uint64_t* term_var, uint32_t* term_coef,
                      uint64_t* var_of_op);
size_t zkgpu_r1cs_coef_bytes(const zkgpu_session* s, uint32_t index, uint8_t* out, size_t cap);
/* A caller-supplied constraint system in CSR form over the session's wire table (retain_all); the term arrays hold
 * row_ptr[3 * n_rows] entries, row_ptr starts at 0 and never decreases, term_coef[i] < n_coefs (all checked).  A term's
 * variable is the index of a value-returning backend call (k < zkgpu_tape_value_ops), an extra variable
 * (k - value_ops < n_extra_vars, stored behind the program's slots) or 0xFFFF...F for the constant one;
 * term_coef indexes coef_bytes (n_coefs little-endian strings of coef_width bytes). */
int zkgpu_r1cs_load_csr(zkgpu_session* s, uint32_t n_rows, const uint32_t* row_ptr, const uint64_t* term_var,
                        const uint32_t* term_coef, const uint8_t* coef_bytes, uint32_t coef_width, uint32_t n_coefs,
                        uint32_t n_extra_vars);
/* rows [first_row, first_row+n_rows): write <a,w>*<b,w> into the single variable of C (witness generation
 * of product rows).  Checked on the host before anything is launched: every row's C is exactly one variable with
 * coefficient 1, no two rows assign the same variable, and no row reads a variable a row of the same call assigns
 * (split the rows by dependency level) -- anything else is an error, never a store to the wrong place. */
int zkgpu_r1cs_assign(zkgpu_session* s, uint32_t first_row, uint32_t n_rows);
int zkgpu_r1cs_check(zkgpu_session* s);                       /* asynchronous: all rows, all lanes */
/* first_fail_row[lane] = smallest violated row or ZKGPU_NO_FAIL; counts = {lanes satisfying all rows, others} */
int zkgpu_r1cs_results(zkgpu_session* s, uint32_t* first_fail_row, uint64_t counts[2]);
/* value of a variable of a loaded CSR for every lane: out[lane][elem_bytes] */
int zkgpu_r1cs_get_var(zkgpu_session* s, uint64_t var, uint8_t* out);
/* the same for n_vars variables at once: out[lane][k][elem_bytes] */
int zkgpu_r1cs_get_vars(zkgpu_session* s, const uint64_t* vars, uint32_t n_vars, uint8_t* out);
/* The quotient ("correction") wires of ToR1CSConverter with use_correction (to_r1cs.rs:163-211,213-260,262-359): for
 * the listed recorded calls (tape indices of add / multiply / add_constant / mul_constant / not calls; the correction
 * variable of call i is var_of_op[i] + 1 of zkgpu_r1cs_export) the integer q = (a op b) / p of every lane, computed on
 * the GPU from the retain_all wire table of the last replay: out[lane][k][elem_bytes], little-endian. */
int zkgpu_r1cs_correction_values(zkgpu_session* s, const uint64_t* tape_ops, uint32_t n_ops, uint8_t* out);
float zkgpu_r1cs_last_ms(const zkgpu_session* s);             /* HIP-event time of the last check */

#ifdef __cplusplus
}
#endif
#endif /* ZKGPU_H */
