"""Python binding of libzkgpu.so (include/zkgpu.h) -- plumbing for tests and bench.py.

The product is the C-ABI library: C++ host (`.sieve` ingest, `Evaluator`,
recording `ZKBackend`, scheduler) + hand-written HIP kernels for gfx950.  This
module only wraps it with ctypes; it contains no evaluation logic and no CPU
fallback -- if the library or a GPU is missing the calls raise.

The directory name contains a hyphen, so import it through `load_package()` in
`__graft_entry__.py` (it registers the module as `zkinterface_ir_amd`).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libzkgpu.so')

NO_FAIL = 0xFFFFFFFF
LANE_NONCANONICAL = 0x1
KIND_NAMES = {1: 'add', 2: 'mul', 3: 'addc', 4: 'mulc', 5: 'copy', 6: 'constant', 7: 'instance', 8: 'witness',
              9: 'assert_zero', 10: 'and', 11: 'xor', 12: 'not'}

_lib = None


class ZkGpuError(RuntimeError):
    pass


def build(force=False):
    """Compile libzkgpu.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(['make', '-s', '-C', _HERE, 'clean'])
    subprocess.check_call(['make', '-s', '-C', _HERE])
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ZkGpuError('libzkgpu.so is not built (run __graft_entry__.build()); there is no fallback path')
    L = ctypes.CDLL(LIB_PATH)
    vp, u8p, u32p, u64p = ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)
    sz, u32, u64, ci = ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
    sig = {
        'zkgpu_session_new': (vp, []),
        'zkgpu_session_free': (None, [vp]),
        'zkgpu_last_error': (ctypes.c_char_p, [vp]),
        'zkgpu_version': (ctypes.c_char_p, []),
        'zkgpu_backend_set_field': (ci, [vp, u8p, sz, u32, ci]),
        'zkgpu_backend_copy': (ci, [vp, u32, u32p]),
        'zkgpu_backend_constant': (ci, [vp, u8p, sz, u32p]),
        'zkgpu_backend_assert_zero': (ci, [vp, u32, u64]),
        'zkgpu_backend_add': (ci, [vp, u32, u32, u32p]),
        'zkgpu_backend_multiply': (ci, [vp, u32, u32, u32p]),
        'zkgpu_backend_add_constant': (ci, [vp, u32, u8p, sz, u32p]),
        'zkgpu_backend_mul_constant': (ci, [vp, u32, u8p, sz, u32p]),
        'zkgpu_backend_and': (ci, [vp, u32, u32, u32p]),
        'zkgpu_backend_xor': (ci, [vp, u32, u32, u32p]),
        'zkgpu_backend_not': (ci, [vp, u32, u32p]),
        'zkgpu_backend_instance': (ci, [vp, u32, u32p]),
        'zkgpu_backend_witness': (ci, [vp, u32, u32p]),
        'zkgpu_backend_drop': (ci, [vp, u32]),
        'zkgpu_stream_info': (ci, [vp, ctypes.POINTER(ctypes.c_double)]),
        'zkgpu_backend_ladder': (ci, [vp, u64, u32, u32]),
        'zkgpu_ingest_messages': (ci, [vp, u8p, sz]),
        'zkgpu_ingest_paths': (ci, [vp, ctypes.POINTER(ctypes.c_char_p), sz]),
        'zkgpu_declare_inputs': (ci, [vp, u32, u32]),
        'zkgpu_host_violations': (sz, [vp, ctypes.c_char_p, sz]),
        'zkgpu_tape_len': (u64, [vp]),
        'zkgpu_tape_value_ops': (u64, [vp]),
        'zkgpu_tape_asserts': (u64, [vp]),
        'zkgpu_tape_dump': (ci, [vp, vp, vp, vp, u64]),
        'zkgpu_tape_assert_wires': (ci, [vp, vp, u64]),
        'zkgpu_n_constants': (u32, [vp]),
        'zkgpu_constant_bytes': (sz, [vp, u32, ctypes.c_char_p, sz]),
        'zkgpu_finalize': (ci, [vp, ci]),
        'zkgpu_elem_bytes': (u32, [vp]),
        'zkgpu_n_instance': (u32, [vp]),
        'zkgpu_n_witness': (u32, [vp]),
        'zkgpu_schedule_info': (ci, [vp, u64p]),
        'zkgpu_schedule_dump': (ci, [vp, vp, vp, vp, vp]),
        'zkgpu_lds_program': (ci, [vp, u32, u64p, vp, vp, vp, vp]),
        'zkgpu_set_inputs': (ci, [vp, vp, vp, u32]),
        'zkgpu_set_inputs_device': (ci, [vp, vp, vp, u32]),
        'zkgpu_set_inputs_from_messages': (ci, [vp]),
        'zkgpu_set_lane_group': (ci, [vp, u32]),
        'zkgpu_input_modes': (sz, [vp, ci, vp, sz]),
        'zkgpu_set_option': (ci, [vp, ctypes.c_char_p, ctypes.c_char_p]),
        'zkgpu_modulus': (sz, [vp, ctypes.c_char_p, sz]),
        'zkgpu_message_values': (u32, [vp, ci]),
        'zkgpu_message_value': (sz, [vp, ci, u32, ctypes.c_char_p, sz]),
        'zkgpu_validator_violations': (sz, [vp, ctypes.c_char_p, sz]),
        'zkgpu_validator_count': (ci, [vp]),
        'zkgpu_validator_live_wires': (ci, [vp]),
        'zkgpu_stats_json': (sz, [vp, ctypes.c_char_p, sz]),
        'zkgpu_stats_warnings': (sz, [vp, ctypes.c_char_p, sz]),
        'zkgpu_uses_lds_path': (ci, [vp]),
        'zkgpu_replay': (ci, [vp]),
        'zkgpu_replay_timed': (ci, [vp]),
        'zkgpu_synchronize': (ci, [vp]),
        'zkgpu_last_replay_ms': (ctypes.c_float, [vp]),
        'zkgpu_launch_timings': (sz, [vp, vp, vp, sz]),
        'zkgpu_counts': (ci, [vp, u64p]),
        'zkgpu_counts_device': (vp, [vp]),
        'zkgpu_stream': (vp, [vp]),
        'zkgpu_n_engines': (ci, [vp]),
        'zkgpu_device_count': (ci, []),
        'zkgpu_n_field_segments': (ci, [vp]),
        'zkgpu_field_segment_info': (ci, [vp, u32, u32p]),
        'zkgpu_field_segment_carried': (ci, [vp, u32, u32p, u32]),
        'zkgpu_field_representation': (ci, [vp, u32]),
        'zkgpu_generic_selftest': (ci, [ctypes.c_char_p, sz, ci, vp, vp, vp, u32p]),
        'zkgpu_rccl_reductions': (u64, [vp]),
        'zkgpu_rccl_note': (sz, [vp, ctypes.c_char_p, sz]),
        'zkgpu_lane_results': (ci, [vp, vp, vp]),
        'zkgpu_lane_violations': (sz, [vp, u32, ctypes.c_char_p, sz]),
        'zkgpu_dump_trace_values': (ci, [vp, u64, u64, vp]),
        'zkgpu_get_wire': (ci, [vp, u64, vp]),
        'zkgpu_table_bytes': (u64, [vp]),
        'zkgpu_r1cs_from_tape': (ci, [vp, ci]),
        'zkgpu_r1cs_info': (ci, [vp, u64p]),
        'zkgpu_schedule_strand_levels': (ctypes.c_size_t, [vp, ctypes.c_uint32, vp, ctypes.c_size_t]),
        'zkgpu_r1cs_class_counts': (ci, [vp, u64p]),
        'zkgpu_r1cs_export': (ci, [vp, vp, vp, vp, vp]),
        'zkgpu_r1cs_coef_bytes': (sz, [vp, u32, ctypes.c_char_p, sz]),
        'zkgpu_r1cs_load_csr': (ci, [vp, u32, vp, vp, vp, vp, u32, u32, u32]),
        'zkgpu_r1cs_assign': (ci, [vp, u32, u32]),
        'zkgpu_r1cs_check': (ci, [vp]),
        'zkgpu_r1cs_results': (ci, [vp, vp, u64p]),
        'zkgpu_r1cs_get_var': (ci, [vp, u64, vp]),
        'zkgpu_r1cs_get_vars': (ci, [vp, vp, u32, vp]),
        'zkgpu_r1cs_correction_values': (ci, [vp, vp, u32, vp]),
        'zkgpu_r1cs_last_ms': (ctypes.c_float, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    L._signatures = sig
    _lib = L
    return L


def exported_symbols():
    return sorted(lib()._signatures.keys())


class Evaluator:
    """Host-side mirror of `consumers::evaluator::Evaluator` bound to the GPU
    tape backend (evaluator.rs:158-303): ingest messages, then evaluate a batch."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.zkgpu_session_new()
        if not self.h:
            raise ZkGpuError('zkgpu_session_new failed')

    def close(self):
        if getattr(self, 'h', None):
            self.L.zkgpu_session_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise ZkGpuError(self.L.zkgpu_last_error(self.h).decode('utf-8', 'replace'))

    # -- Evaluator::from_messages / ingest_message ------------------------------
    def ingest_message(self, data):
        data = bytes(data)
        self._ck(self.L.zkgpu_ingest_messages(self.h, data, len(data)))

    def ingest_paths(self, paths):
        arr = (ctypes.c_char_p * len(paths))(*[p.encode() for p in paths])
        self._ck(self.L.zkgpu_ingest_paths(self.h, arr, len(paths)))

    @classmethod
    def from_messages(cls, buffers, **options):
        ev = cls()
        for k, v in options.items():
            ev.set_option(k, str(v))
        for b in buffers:
            ev.ingest_message(b)
        return ev

    def declare_inputs(self, n_instance, n_witness):
        self._ck(self.L.zkgpu_declare_inputs(self.h, n_instance, n_witness))

    def _text(self, fn, *args):
        n = fn(self.h, *args, None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        fn(self.h, *args, buf, n + 1)
        return buf.value.decode('utf-8', 'replace')

    def host_violations(self):
        s = self._text(self.L.zkgpu_host_violations)
        return s.split('\n') if s else []

    def modulus_le(self):
        n = self.L.zkgpu_modulus(self.h, None, 0)
        buf = ctypes.create_string_buffer(max(n, 1))
        self.L.zkgpu_modulus(self.h, buf, n)
        return buf.raw[:n]

    def message_values(self, witness=False):
        """Values of the ingested Instance / Witness messages in stream order (little-endian bytes as sent)."""
        out = []
        for k in range(self.L.zkgpu_message_values(self.h, int(witness))):
            n = self.L.zkgpu_message_value(self.h, int(witness), k, None, 0)
            buf = ctypes.create_string_buffer(max(n, 1))
            self.L.zkgpu_message_value(self.h, int(witness), k, buf, n)
            out.append(buf.raw[:n])
        return out

    # -- the other two consumers of `valid-eval-metrics` (cli.rs:333-363) -------
    def validator_violations(self):
        """Validator::get_violations() (validator.rs:135-142); needs set_option('validate', 'prover'|'verifier')
        before the first message."""
        if self.L.zkgpu_validator_count(self.h) < 0:
            raise ZkGpuError('the validator is not enabled (set_option("validate", "prover") before ingesting)')
        s = self._text(self.L.zkgpu_validator_violations)
        return s.split('\n') if s else []

    def validator_has_live_wires(self):
        return self.L.zkgpu_validator_live_wires(self.h) == 1

    def stats_json(self):
        """serde_json::to_writer_pretty(&Stats) (stats.rs:44-53, cli.rs:353); needs set_option('metrics', '1')."""
        return self._text(self.L.zkgpu_stats_json)

    def stats(self):
        import json
        return json.loads(self.stats_json())

    def stats_warnings(self):
        s = self._text(self.L.zkgpu_stats_warnings)
        return s.split('\n') if s else []

    # -- ZKBackend trait methods (evaluator.rs:17-76), one C entry point each ------
    def _wire(self, fn, *args):
        out = ctypes.c_uint32(0)
        self._ck(fn(self.h, *args, ctypes.byref(out)))
        return out.value

    def backend_set_field(self, modulus_le, degree=1, is_boolean=False):
        m = bytes(modulus_le)
        self._ck(self.L.zkgpu_backend_set_field(self.h, m, len(m), degree, 1 if is_boolean else 0))

    def backend_copy(self, w):
        return self._wire(self.L.zkgpu_backend_copy, w)

    def backend_constant(self, value_le):
        v = bytes(value_le)
        return self._wire(self.L.zkgpu_backend_constant, v, len(v))

    def backend_assert_zero(self, w, local_wire_id=0):
        self._ck(self.L.zkgpu_backend_assert_zero(self.h, w, local_wire_id))

    def backend_add(self, a, b):
        return self._wire(self.L.zkgpu_backend_add, a, b)

    def backend_multiply(self, a, b):
        return self._wire(self.L.zkgpu_backend_multiply, a, b)

    def backend_add_constant(self, a, c_le):
        c = bytes(c_le)
        return self._wire(self.L.zkgpu_backend_add_constant, a, c, len(c))

    def backend_mul_constant(self, a, c_le):
        c = bytes(c_le)
        return self._wire(self.L.zkgpu_backend_mul_constant, a, c, len(c))

    def backend_and(self, a, b):
        return self._wire(self.L.zkgpu_backend_and, a, b)

    def backend_xor(self, a, b):
        return self._wire(self.L.zkgpu_backend_xor, a, b)

    def backend_not(self, a):
        return self._wire(self.L.zkgpu_backend_not, a)

    def backend_instance(self, position):
        return self._wire(self.L.zkgpu_backend_instance, position)

    def backend_ladder(self, first_call, base, result):
        """hint: the calls first_call.. computed result = base^(modulus - 1) (include/zkgpu.h)"""
        self._ck(self.L.zkgpu_backend_ladder(self.h, first_call, base, result))

    def backend_drop(self, w):
        self._ck(self.L.zkgpu_backend_drop(self.h, w))

    def backend_witness(self, position):
        return self._wire(self.L.zkgpu_backend_witness, position)

    # -- tape --------------------------------------------------------------------
    def tape(self):
        import numpy as np
        n = self.L.zkgpu_tape_len(self.h)
        kinds = np.zeros(n, dtype=np.uint8)
        a = np.zeros(n, dtype=np.uint32)
        b = np.zeros(n, dtype=np.uint32)
        if n:
            self._ck(self.L.zkgpu_tape_dump(self.h, kinds.ctypes.data, a.ctypes.data, b.ctypes.data, n))
        return kinds, a, b

    def assert_wires(self):
        import numpy as np
        n = self.n_asserts
        out = np.zeros(max(n, 1), dtype=np.uint64)
        self._ck(self.L.zkgpu_tape_assert_wires(self.h, out.ctypes.data, n))
        return out[:n]

    @property
    def n_constants(self):
        """constants of the (inspected) tape; the pool schedule_dump returns holds, behind their device forms, the raw
        integers of those whose unreduced bits are read (include/zkgpu.h zkgpu_schedule_dump)"""
        return int(self.L.zkgpu_n_constants(self.h))

    def constants(self):
        out = []
        for i in range(self.L.zkgpu_n_constants(self.h)):
            n = self.L.zkgpu_constant_bytes(self.h, i, None, 0)
            buf = ctypes.create_string_buffer(max(n, 1))
            self.L.zkgpu_constant_bytes(self.h, i, buf, n)
            out.append(buf.raw[:n])
        return out

    @property
    def tape_len(self):
        """backend calls recorded so far (value-returning calls and assert_zero)"""
        return self.L.zkgpu_tape_len(self.h)

    @property
    def n_value_ops(self):
        return self.L.zkgpu_tape_value_ops(self.h)

    @property
    def n_asserts(self):
        return self.L.zkgpu_tape_asserts(self.h)

    # -- batch replay -------------------------------------------------------------
    def finalize(self, retain_all=False):
        self._ck(self.L.zkgpu_finalize(self.h, 1 if retain_all else 0))

    @property
    def elem_bytes(self):
        return self.L.zkgpu_elem_bytes(self.h)

    @property
    def n_instance(self):
        return self.L.zkgpu_n_instance(self.h)

    @property
    def n_witness(self):
        return self.L.zkgpu_n_witness(self.h)

    def stream_info(self):
        out = (ctypes.c_double * 3)()
        self._ck(self.L.zkgpu_stream_info(self.h, out))
        return {'windows': int(out[0]), 'streamed_windows': int(out[1]), 'worker_busy_s': float(out[2])}

    def schedule_info(self):
        out = (ctypes.c_uint64 * 8)()
        self._ck(self.L.zkgpu_schedule_info(self.h, out))
        keys = ['levels', 'launches', 'slots', 'max_width', 'sequential_launches', 'device_ops', 'const_words',
                'words_per_const']
        return dict(zip(keys, list(out)))

    def strand_levels(self, launch):
        """(level bounds relative to the launch's first entry, LDS-resident values) of a strand, None for any other launch
        (include/zkgpu.h zkgpu_schedule_strand_levels)"""
        import numpy as np
        n = self.L.zkgpu_schedule_strand_levels(self.h, launch, None, 0)
        if not n:
            return None
        out = np.zeros(n, dtype=np.uint32)
        self.L.zkgpu_schedule_strand_levels(self.h, launch, out.ctypes.data, n)
        return out[:-1].copy(), int(out[-1])

    def schedule_dump(self):
        import numpy as np
        info = self.schedule_info()
        ops = np.zeros((info['device_ops'], 8), dtype=np.uint32)
        launches = np.zeros((info['launches'], 4), dtype=np.uint32)
        consts = np.zeros(max(info['const_words'], 1), dtype=np.uint32)
        slot_of = np.zeros(max(self.L.zkgpu_tape_len(self.h), 1), dtype=np.uint32)
        self._ck(self.L.zkgpu_schedule_dump(self.h, ops.ctypes.data, launches.ctypes.data, consts.ctypes.data,
                                            slot_of.ctypes.data))
        return ops, launches, consts[:info['const_words']], slot_of[:self.L.zkgpu_tape_len(self.h)]

    def lds_program(self, block_rows=0):
        """GF(2): the program of the LDS-resident kernel for this schedule (host work, no GPU): dict with `ops8`
        [n][4] u16 {dst, a, b, kind}, `rows` u16 stream, `blocks` [n][2] u32, `chunks` [n][4] u32, `block_rows`,
        `table_words` (csrc/device/lds_layout.hpp)."""
        import numpy as np
        sizes = (ctypes.c_uint64 * 6)()
        self._ck(self.L.zkgpu_lds_program(self.h, block_rows, sizes, None, None, None, None))
        n_ops, n_rows, n_blocks, n_chunks, br, words = [int(x) for x in sizes]
        ops8 = np.zeros((max(n_ops, 1), 4), dtype=np.uint16)
        rows = np.zeros(max(n_rows, 1), dtype=np.uint16)
        blocks = np.zeros((max(n_blocks, 1), 2), dtype=np.uint32)
        chunks = np.zeros((max(n_chunks, 1), 4), dtype=np.uint32)
        self._ck(self.L.zkgpu_lds_program(self.h, block_rows, sizes, ops8.ctypes.data, rows.ctypes.data, blocks.ctypes.data,
                                          chunks.ctypes.data))
        return {'ops8': ops8[:n_ops], 'rows': rows[:n_rows], 'blocks': blocks[:n_blocks], 'chunks': chunks[:n_chunks],
                'block_rows': br, 'table_words': words}

    def set_inputs(self, instances, witnesses, batch):
        """instances / witnesses: bytes-like of [batch][n][elem_bytes] (or None when n == 0), or the integer address
        of such a host buffer (e.g. `tensor.data_ptr()` of a pinned torch tensor: read by DMA without a staging copy)."""
        def arg(x):
            if x is None or isinstance(x, int):
                return x
            return bytes(x)
        self._keep = (arg(instances), arg(witnesses))
        self._ck(self.L.zkgpu_set_inputs(self.h, self._keep[0], self._keep[1], batch))

    def set_inputs_device(self, d_instances, d_witnesses, batch):
        self._ck(self.L.zkgpu_set_inputs_device(self.h, d_instances, d_witnesses, batch))

    def set_inputs_from_messages(self):
        self._ck(self.L.zkgpu_set_inputs_from_messages(self.h))

    @property
    def n_field_segments(self):
        """parts of the relation recorded under one field characteristic each (include/zkgpu.h "Field segments")"""
        return int(self.L.zkgpu_n_field_segments(self.h))

    def field_segment_info(self, k):
        out = (ctypes.c_uint32 * 4)()
        self._ck(self.L.zkgpu_field_segment_info(self.h, k, out))
        return dict(zip(['carried_in', 'assert_base', 'words', 'carried_out'], [int(x) for x in out]))

    def field_segment_carried(self, k):
        """wire-table slots of the values segment k hands to segment k + 1, in carry order"""
        n = self.field_segment_info(k)['carried_out']
        out = (ctypes.c_uint32 * max(n, 1))()
        self._ck(self.L.zkgpu_field_segment_carried(self.h, k, out, n))
        return [int(x) for x in out[:n]]

    def field_representation(self, k=0):
        """0 = bit-packed GF(2), 1 = Montgomery form, 2 = canonical residues (the any-modulus kernels)"""
        return int(self.L.zkgpu_field_representation(self.h, k))

    def input_modes(self, witness=False):
        """per input position how a value >= p is treated (include/zkgpu.h zkgpu_input_modes): list of 0x00 / 0x01 / 0x02 / 0xFF
        (witness = 2: the values carried into the inspected field segment)"""
        n = self.L.zkgpu_input_modes(self.h, int(witness), None, 0)
        buf = (ctypes.c_uint8 * max(n, 1))()
        self.L.zkgpu_input_modes(self.h, int(witness), buf, n)
        return list(buf[:n])

    def set_lane_group(self, lanes):
        self._ck(self.L.zkgpu_set_lane_group(self.h, lanes))

    def set_option(self, key, value):
        self._ck(self.L.zkgpu_set_option(self.h, key.encode(), value.encode()))

    def uses_lds_path(self):
        r = self.L.zkgpu_uses_lds_path(self.h)
        if r < 0:
            self._ck(1)
        return bool(r)

    def replay(self):
        self._ck(self.L.zkgpu_replay(self.h))

    def replay_timed(self):
        import numpy as np
        self._ck(self.L.zkgpu_replay_timed(self.h))
        n = self.L.zkgpu_launch_timings(self.h, None, None, 0)
        ms = np.zeros(n, dtype=np.float32)
        ops = np.zeros(n, dtype=np.uint32)
        self.L.zkgpu_launch_timings(self.h, ms.ctypes.data, ops.ctypes.data, n)
        return ms, ops

    def synchronize(self):
        self._ck(self.L.zkgpu_synchronize(self.h))

    @property
    def last_replay_ms(self):
        return float(self.L.zkgpu_last_replay_ms(self.h))

    def counts(self):
        out = (ctypes.c_uint64 * 2)()
        self._ck(self.L.zkgpu_counts(self.h, out))
        return int(out[0]), int(out[1])

    def counts_device_ptr(self):
        return self.L.zkgpu_counts_device(self.h)

    @property
    def n_engines(self):
        return int(self.L.zkgpu_n_engines(self.h))

    @property
    def rccl_reductions(self):
        """counts() calls answered by an RCCL all-reduce (several devices, or option force_rccl)"""
        return int(self.L.zkgpu_rccl_reductions(self.h))

    def rccl_note(self):
        return self._text(self.L.zkgpu_rccl_note)

    def stream_ptr(self):
        return self.L.zkgpu_stream(self.h)

    def lane_results(self, batch):
        import numpy as np
        ff = np.zeros(batch, dtype=np.uint32)
        fl = np.zeros(batch, dtype=np.uint32)
        self._ck(self.L.zkgpu_lane_results(self.h, ff.ctypes.data, fl.ctypes.data))
        return ff, fl

    def get_violations(self, lane=0):
        """`Evaluator::get_violations()` (evaluator.rs:199-208) for one lane."""
        n = self.L.zkgpu_lane_violations(self.h, lane, None, 0)
        buf = ctypes.create_string_buffer(n + 1)
        self.L.zkgpu_lane_violations(self.h, lane, buf, n + 1)
        s = buf.value.decode('utf-8', 'replace')
        return s.split('\n') if s else []

    def dump_trace_values(self, batch, first=0, count=None):
        """[batch][count] python ints: value of every value-returning backend call (retain_all only)."""
        if count is None:
            count = self.n_value_ops - first
        w = self.elem_bytes
        buf = ctypes.create_string_buffer(max(batch * count * w, 1))
        self._ck(self.L.zkgpu_dump_trace_values(self.h, first, count, buf))
        raw = buf.raw
        return [[int.from_bytes(raw[(l * count + k) * w:(l * count + k + 1) * w], 'little') for k in range(count)]
                for l in range(batch)]

    def get(self, wire_id, batch):
        """`Evaluator::get` (evaluator.rs:750-752): value of a live top-level wire, per lane."""
        w = self.elem_bytes
        buf = ctypes.create_string_buffer(max(batch * w, 1))
        rc = self.L.zkgpu_get_wire(self.h, wire_id, buf)
        if rc == 3:
            return None
        self._ck(rc)
        return [int.from_bytes(buf.raw[l * w:(l + 1) * w], 'little') for l in range(batch)]

    # -- R1CS (ir-to-zkif) -----------------------------------------------------------
    def r1cs_from_tape(self, use_correction=False):
        self._ck(self.L.zkgpu_r1cs_from_tape(self.h, 1 if use_correction else 0))

    def r1cs_info(self):
        out = (ctypes.c_uint64 * 4)()
        self._ck(self.L.zkgpu_r1cs_info(self.h, out))
        return dict(zip(['rows', 'vars', 'terms', 'coefs'], list(out)))

    def r1cs_class_counts(self):
        """combinations of the rows by coefficient class (include/zkgpu.h zkgpu_r1cs_class_counts)"""
        out = (ctypes.c_uint64 * 3)()
        self._ck(self.L.zkgpu_r1cs_class_counts(self.h, out))
        return dict(zip(['full', 'unit', 'small'], [int(x) for x in out]))

    def r1cs_export(self):
        """(rows, var_of_op): rows = list of (A, B, C), each a list of (variable, coefficient int)"""
        import numpy as np
        info = self.r1cs_info()
        row_ptr = np.zeros(3 * info['rows'] + 1, dtype=np.uint32)
        tv = np.zeros(max(info['terms'], 1), dtype=np.uint64)
        tc = np.zeros(max(info['terms'], 1), dtype=np.uint32)
        vo = np.zeros(max(self.L.zkgpu_tape_len(self.h), 1), dtype=np.uint64)
        self._ck(self.L.zkgpu_r1cs_export(self.h, row_ptr.ctypes.data, tv.ctypes.data, tc.ctypes.data, vo.ctypes.data))
        coefs = []
        for i in range(info['coefs']):
            n = self.L.zkgpu_r1cs_coef_bytes(self.h, i, None, 0)
            buf = ctypes.create_string_buffer(max(n, 1))
            self.L.zkgpu_r1cs_coef_bytes(self.h, i, buf, n)
            coefs.append(int.from_bytes(buf.raw[:n], 'little'))
        rows = []
        for r in range(info['rows']):
            parts = []
            for k in range(3):
                t0, t1 = int(row_ptr[3 * r + k]), int(row_ptr[3 * r + k + 1])
                parts.append([(int(tv[t]), coefs[int(tc[t])]) for t in range(t0, t1)])
            rows.append(tuple(parts))
        return rows, vo[:self.L.zkgpu_tape_len(self.h)]

    def r1cs_load_csr(self, row_ptr, term_var, term_coef, coef_bytes, coef_width, n_extra_vars):
        import numpy as np
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        term_var = np.ascontiguousarray(term_var, dtype=np.uint64)
        term_coef = np.ascontiguousarray(term_coef, dtype=np.uint32)
        coef_bytes = np.ascontiguousarray(coef_bytes, dtype=np.uint8)
        n_rows = (len(row_ptr) - 1) // 3
        self._ck(self.L.zkgpu_r1cs_load_csr(self.h, n_rows, row_ptr.ctypes.data, term_var.ctypes.data,
                                            term_coef.ctypes.data, coef_bytes.ctypes.data, coef_width,
                                            coef_bytes.size // coef_width, n_extra_vars))

    def r1cs_assign(self, first_row, n_rows):
        self._ck(self.L.zkgpu_r1cs_assign(self.h, first_row, n_rows))

    def r1cs_check(self):
        self._ck(self.L.zkgpu_r1cs_check(self.h))

    def r1cs_results(self, batch):
        import numpy as np
        ff = np.zeros(batch, dtype=np.uint32)
        counts = (ctypes.c_uint64 * 2)()
        self._ck(self.L.zkgpu_r1cs_results(self.h, ff.ctypes.data, counts))
        return ff, (int(counts[0]), int(counts[1]))

    def r1cs_get_var(self, var, batch):
        w = self.elem_bytes
        buf = ctypes.create_string_buffer(max(batch * w, 1))
        self._ck(self.L.zkgpu_r1cs_get_var(self.h, var, buf))
        return [int.from_bytes(buf.raw[l * w:(l + 1) * w], 'little') for l in range(batch)]

    def r1cs_correction_values(self, tape_ops, batch):
        """q[lane][k] = the integer quotient (a op b) // p of the k-th listed call (to_r1cs.rs use_correction)"""
        import numpy as np
        w = self.elem_bytes
        ids = np.ascontiguousarray(tape_ops, dtype=np.uint64)
        buf = ctypes.create_string_buffer(max(batch * len(ids) * w, 1))
        self._ck(self.L.zkgpu_r1cs_correction_values(self.h, ids.ctypes.data, len(ids), buf))
        raw = buf.raw
        return [[int.from_bytes(raw[(l * len(ids) + k) * w:(l * len(ids) + k + 1) * w], 'little') for k in range(len(ids))]
                for l in range(batch)]

    def r1cs_get_vars(self, variables, batch):
        """values[lane][k] of the k-th listed variable (one device dump for all of them)"""
        import numpy as np
        w = self.elem_bytes
        ids = np.ascontiguousarray(variables, dtype=np.uint64)
        buf = ctypes.create_string_buffer(max(batch * len(ids) * w, 1))
        self._ck(self.L.zkgpu_r1cs_get_vars(self.h, ids.ctypes.data, len(ids), buf))
        raw = buf.raw
        return [[int.from_bytes(raw[(l * len(ids) + k) * w:(l * len(ids) + k + 1) * w], 'little') for k in range(len(ids))]
                for l in range(batch)]

    @property
    def r1cs_last_ms(self):
        return float(self.L.zkgpu_r1cs_last_ms(self.h))

    @property
    def table_bytes(self):
        return self.L.zkgpu_table_bytes(self.h)


def generic_selftest(p, op, a, b=0):
    """Test hook (include/zkgpu.h zkgpu_generic_selftest): the any-modulus kernels' arithmetic on the host, Python integers
    in and out.  op: 'add', 'mul', 'reduce', 'and', 'xor'."""
    L = lib()
    mod = int(p).to_bytes((int(p).bit_length() + 7) // 8 or 1, 'little')
    nw = ctypes.c_uint32(0)
    rc = L.zkgpu_generic_selftest(mod, len(mod), 0, None, None, None, ctypes.byref(nw))
    if rc:
        raise ZkGpuError('zkgpu_generic_selftest: the modulus is not supported (%d)' % rc)
    n = nw.value
    A = (ctypes.c_uint32 * n)(*[(int(a) >> (32 * i)) & 0xFFFFFFFF for i in range(n)])
    B = (ctypes.c_uint32 * n)(*[(int(b) >> (32 * i)) & 0xFFFFFFFF for i in range(n)])
    out = (ctypes.c_uint32 * n)()
    rc = L.zkgpu_generic_selftest(mod, len(mod), {'add': 0, 'mul': 1, 'reduce': 2, 'and': 3, 'xor': 4}[op], A, B, out, ctypes.byref(nw))
    if rc:
        raise ZkGpuError('zkgpu_generic_selftest failed (%d)' % rc)
    return sum(int(out[i]) << (32 * i) for i in range(n))


def _ingest_any(ev, paths_or_buffers):
    if paths_or_buffers and isinstance(paths_or_buffers[0], str):
        ev.ingest_paths(list(paths_or_buffers))
    else:
        for b in paths_or_buffers:
            ev.ingest_message(b)


def validate(paths_or_buffers, as_prover=True):
    """`zki_sieve validate` (cli.rs:302-313): the Validator's violation list, no GPU involved."""
    ev = Evaluator()
    ev.set_option('validate', 'prover' if as_prover else 'verifier')
    _ingest_any(ev, paths_or_buffers)
    return ev.validator_violations()


def metrics(paths_or_buffers):
    """`zki_sieve metrics` (cli.rs:322-330): the Stats JSON text, no GPU involved."""
    ev = Evaluator()
    ev.set_option('metrics', '1')
    _ingest_any(ev, paths_or_buffers)
    return ev.stats_json()


def valid_eval_metrics(paths_or_buffers, stream=True):
    """`zki_sieve valid-eval-metrics` (cli.rs:333-363): every message is read once and fed to the Validator
    (as prover), the Evaluator and the Stats; returns (validator violations, evaluator violations, stats JSON).
    The evaluation itself is the GPU replay of `evaluate`."""
    ev = Evaluator()
    ev.set_option('validate', 'prover')
    ev.set_option('metrics', '1')
    ev.set_option('stream', '1' if stream else '0')
    _ingest_any(ev, paths_or_buffers)
    return ev.validator_violations(), _finish_evaluate(ev), ev.stats_json()


def evaluate(paths_or_buffers, stream=True):
    """`zki_sieve evaluate` (cli.rs:315-320) for one statement: returns the violation list
    (empty list = "The statement is TRUE!").  stream: one statement, one witness -- what counts is relation in -> verdict
    out, so the windows of the tape are scheduled (and, GF(2), the LDS program built) by the worker thread while the rest
    of the relation is still being read (option "stream"; the verdict is the same either way, tests/test_stream.py)."""
    ev = Evaluator()
    ev.set_option('stream', '1' if stream else '0')
    _ingest_any(ev, paths_or_buffers)
    return _finish_evaluate(ev)


def _finish_evaluate(ev):
    try:
        ev.finalize()
    except ZkGpuError:
        # no Relation reached the backend: only the recording-time violations exist
        return ev.host_violations()
    ev.set_inputs_from_messages()
    ev.replay()
    ev.synchronize()
    return ev.get_violations(0)
