"""ctypes binding of the parity oracle (oracle/libzkoracle.so).  Test
infrastructure only: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package."""
import ctypes
import hashlib
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(_ROOT, 'oracle', 'libzkoracle.so')

TRACE_NAMES = ['copy', 'constant', 'add', 'mul', 'addc', 'mulc', 'and', 'xor', 'not', 'instance', 'witness']


def build():
    subprocess.check_call(['make', '-s', '-C', os.path.join(_ROOT, 'oracle')])


def load():
    if not os.path.exists(_LIB):
        build()
    lib = ctypes.CDLL(_LIB)
    lib.zko_new.restype = ctypes.c_void_p
    lib.zko_new.argtypes = [ctypes.c_int]
    lib.zko_free.argtypes = [ctypes.c_void_p]
    lib.zko_set_max_ops.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    lib.zko_ingest_buffer.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    lib.zko_ingest_files.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_int]
    lib.zko_n_violations.argtypes = [ctypes.c_void_p]
    lib.zko_violation.restype = ctypes.c_char_p
    lib.zko_violation.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.zko_panicked.argtypes = [ctypes.c_void_p]
    for f in ('zko_n_ops', 'zko_n_asserts', 'zko_trace_len', 'zko_n_live_wires'):
        getattr(lib, f).restype = ctypes.c_uint64
        getattr(lib, f).argtypes = [ctypes.c_void_p]
    lib.zko_queue_len.restype = ctypes.c_uint64
    lib.zko_queue_len.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.zko_trace_kinds.restype = ctypes.POINTER(ctypes.c_uint8)
    lib.zko_trace_kinds.argtypes = [ctypes.c_void_p]
    lib.zko_trace_text.restype = ctypes.c_char_p
    lib.zko_trace_text.argtypes = [ctypes.c_void_p]
    lib.zko_trace_values_le.restype = ctypes.c_uint64
    lib.zko_trace_values_le.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint32]
    lib.zko_get_wire_le.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint32]
    lib.zko_exp.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p,
                            ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32]
    lib.zko_eval_batch.restype = ctypes.c_double
    lib.zko_eval_batch.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32,
                                   ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32,
                                   ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p,
                                   ctypes.POINTER(ctypes.c_uint64)]
    lib.zko_opt_eval.restype = ctypes.c_double
    lib.zko_opt_eval.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_char_p,
                                 ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_char_p,
                                 ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                 ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    lib.zko_opt_eval_dump.restype = ctypes.c_double
    lib.zko_opt_eval_dump.argtypes = lib.zko_opt_eval.argtypes + [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p]
    lib.zko_r1cs_check.restype = ctypes.c_double
    lib.zko_r1cs_check.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p,
                                   ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p,
                                   ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                   ctypes.c_uint32, ctypes.c_void_p]
    return lib


def r1cs_check(row_ptr, term_var, term_coef, coef_bytes, modulus_le, base, n_vars, n_assign, threads):
    """CPU row check of a CSR system (oracle/cpu_opt.cpp zko_r1cs_check): base = uint8 [batch][n_base][width] values of
    the variables 0..n_base-1, the rest are assigned by the first n_assign product rows.  Returns (first_fail, seconds)."""
    import numpy as np
    lib = load()
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
    term_var = np.ascontiguousarray(term_var, dtype=np.uint64)
    term_coef = np.ascontiguousarray(term_coef, dtype=np.uint32)
    coef_bytes = np.ascontiguousarray(coef_bytes, dtype=np.uint8)
    base = np.ascontiguousarray(base, dtype=np.uint8)
    batch, n_base, width = base.shape
    ff = np.zeros(batch, dtype=np.uint32)
    secs = lib.zko_r1cs_check(row_ptr.ctypes.data, term_var.ctypes.data, term_coef.ctypes.data, (len(row_ptr) - 1) // 3,
                              coef_bytes.tobytes(), coef_bytes.shape[1], coef_bytes.shape[0], modulus_le, len(modulus_le),
                              base.ctypes.data, n_base, n_vars, width, n_assign, batch, threads, ff.ctypes.data)
    return ff, secs


def opt_eval(kinds, a, b, constants, modulus_le, inst, n_inst, wit, n_wit, width, batch, threads, dump_lane=None):
    """`cpu_opt`: flat-array Montgomery evaluation of a recorded tape (oracle/cpu_opt.cpp).
    Returns (first_fail uint32[batch], seconds, values of dump_lane or None)."""
    import numpy as np
    lib = load()
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    a = np.ascontiguousarray(a, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.uint32)
    cw = max([len(c) for c in constants] + [1])
    cbytes = b''.join(bytes(c) + bytes(cw - len(c)) for c in constants) or b'\x00'
    ff = np.zeros(batch, dtype=np.uint32)
    vals = np.zeros((len(kinds), 32), dtype=np.uint8) if dump_lane is not None else None
    secs = lib.zko_opt_eval(kinds.ctypes.data, a.ctypes.data, b.ctypes.data, len(kinds), cbytes, cw, len(constants),
                            modulus_le, len(modulus_le), inst, n_inst, wit, n_wit, width, batch, threads,
                            ff.ctypes.data, vals.ctypes.data if vals is not None else None, dump_lane or 0)
    out = None
    if vals is not None:
        out = [int.from_bytes(vals[i].tobytes(), 'little') for i in range(len(kinds)) if kinds[i] != 9]
    return ff, secs, out


def opt_eval_outputs(kinds, a, b, constants, modulus_le, inst, n_inst, wit, n_wit, width, batch, threads, dump_ops):
    """`cpu_opt` over a whole batch, returning for every lane the canonical values of the listed tape ops:
    (first_fail uint32[batch], seconds, uint8 [batch][len(dump_ops)][32] little-endian)."""
    import numpy as np
    lib = load()
    kinds = np.ascontiguousarray(kinds, dtype=np.uint8)
    a = np.ascontiguousarray(a, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.uint32)
    cw = max([len(c) for c in constants] + [1])
    cbytes = b''.join(bytes(c) + bytes(cw - len(c)) for c in constants) or b'\x00'
    ff = np.zeros(batch, dtype=np.uint32)
    dump_ops = np.ascontiguousarray(dump_ops, dtype=np.uint64)
    assert len(dump_ops) and int(dump_ops.max()) < len(kinds)
    out = np.zeros((batch, len(dump_ops), 32), dtype=np.uint8)
    secs = lib.zko_opt_eval_dump(kinds.ctypes.data, a.ctypes.data, b.ctypes.data, len(kinds), cbytes, cw, len(constants),
                                 modulus_le, len(modulus_le), inst, n_inst, wit, n_wit, width, batch, threads,
                                 ff.ctypes.data, None, 0, dump_ops.ctypes.data, len(dump_ops), out.ctypes.data)
    return ff, secs, out


class OracleRun:
    """One reference-Evaluator run over a sequence of message buffers."""

    def __init__(self, buffers=None, files=None, trace=True, width=32, max_ops=None):
        self.lib = load()
        self.h = self.lib.zko_new(1 if trace else 0)
        if max_ops:
            self.lib.zko_set_max_ops(self.h, max_ops)
        self.width = width
        if files is not None:
            arr = (ctypes.c_char_p * len(files))(*[f.encode() for f in files])
            self.lib.zko_ingest_files(self.h, arr, len(files))
        for b in buffers or []:
            self.lib.zko_ingest_buffer(self.h, bytes(b), len(b))

    def __del__(self):
        try:
            self.lib.zko_free(self.h)
        except Exception:
            pass

    @property
    def violations(self):
        n = self.lib.zko_n_violations(self.h)
        return [self.lib.zko_violation(self.h, i).decode('utf-8', 'replace') for i in range(n)]

    @property
    def panicked(self):
        return bool(self.lib.zko_panicked(self.h))

    @property
    def n_ops(self):
        return self.lib.zko_n_ops(self.h)

    @property
    def n_asserts(self):
        return self.lib.zko_n_asserts(self.h)

    def trace_kinds(self):
        n = self.lib.zko_trace_len(self.h)
        p = self.lib.zko_trace_kinds(self.h)
        return [TRACE_NAMES[p[i]] for i in range(n)]

    def trace_text(self):
        return self.lib.zko_trace_text(self.h).decode()

    def trace_sha256(self):
        return hashlib.sha256(self.trace_text().encode()).hexdigest()

    def trace_values(self):
        n = self.lib.zko_trace_len(self.h)
        buf = ctypes.create_string_buffer(n * self.width)
        bad = self.lib.zko_trace_values_le(self.h, buf, self.width)
        assert bad == 0, 'values wider than %d bytes' % self.width
        raw = buf.raw
        return [int.from_bytes(raw[i * self.width:(i + 1) * self.width], 'little') for i in range(n)]

    def get(self, wire_id):
        buf = ctypes.create_string_buffer(self.width)
        r = self.lib.zko_get_wire_le(self.h, wire_id, buf, self.width)
        if r == 0:
            return None
        assert r == 1
        return int.from_bytes(buf.raw, 'little')

    def n_live_wires(self):
        return self.lib.zko_n_live_wires(self.h)

    def queue_len(self, which):
        return self.lib.zko_queue_len(self.h, which)


def oracle_exp(base, exponent, modulus, width=32):
    lib = load()

    def le(v):
        return v.to_bytes(max(1, (v.bit_length() + 7) // 8), 'little')
    out = ctypes.create_string_buffer(width)
    b, e, m = le(base), le(exponent), le(modulus)
    rc = lib.zko_exp(b, len(b), e, len(e), m, len(m), out, width)
    assert rc == 0
    return int.from_bytes(out.raw, 'little')


def eval_batch(relation_bytes, modulus_le, inst, n_inst, wit, n_wit, width, batch, threads):
    """inst/wit: bytes of [batch][n][width].  Returns (ok list, seconds, total backend ops)."""
    lib = load()
    ok = ctypes.create_string_buffer(batch)
    ops = ctypes.c_uint64(0)
    secs = lib.zko_eval_batch(relation_bytes, len(relation_bytes), modulus_le, len(modulus_le), inst, n_inst, wit,
                              n_wit, width, batch, threads, ok, ctypes.byref(ops))
    return list(ok.raw), secs, ops.value
