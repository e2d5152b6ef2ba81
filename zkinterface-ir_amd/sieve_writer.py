"""FlatBuffers writer for SIEVE IR messages (`.sieve`): emits synthetic
workspaces (bench.py workloads) and the circuits used by the tests.

Builds size-prefixed buffers following /root/reference/sieve_ir.fbs (vtable
slots = the VT_* constants of rust/src/sieve_ir_generated.rs) so
that circuits can be stated as *data* (tuples) and emitted as real workspaces
the original `zki_sieve` could consume (write half of rust/src/structs/*.rs,
sinks of rust/src/producers/sink.rs:84-100).  Not a translation of the
reference's builders: a small back-to-front table writer plus a numpy bulk
path for million-gate relations.

Gate tuples (mirroring rust/src/structs/gates.rs:17-55):
  ('constant', out, bytes) ('assert_zero', inp) ('copy', out, inp)
  ('add'|'mul'|'and'|'xor', out, l, r) ('addc'|'mulc', out, inp, bytes)
  ('not', out, inp) ('instance', out) ('witness', out) ('free', first, last|None)
  ('call', name, outs, ins) ('anoncall', outs, ins, n_inst, n_wit, [gates])
  ('switch', cond, outs, [case bytes], [branch]) with
      branch = ('call', name, ins) | ('anon', ins, n_inst, n_wit, [gates])
  ('for', iterator, first, last, outs, body) with
      body = ('call', name, out_iterlist, in_iterlist)
           | ('anon', out_iterlist, in_iterlist, n_inst, n_wit, [gates])
wirelist  = [int | (first, last)]
iterexpr  = ('const', v) | ('name', s) | ('add'|'sub'|'mul', l, r) | ('div', numer, denom)
iterlist  = [iterexpr | ('range', first, last)]
"""
import struct

import numpy as np

DIRECTIVE_TAGS = {
    'constant': 1, 'assert_zero': 2, 'copy': 3, 'add': 4, 'mul': 5, 'addc': 6, 'mulc': 7,
    'and': 8, 'xor': 9, 'not': 10, 'instance': 11, 'witness': 12, 'free': 13, 'call': 14,
    'anoncall': 15, 'switch': 16, 'for': 17,
}


class Builder:
    """Back-to-front FlatBuffers writer.  Positions are `end offsets`: the
    distance from the end of the finished buffer to the start of the object."""

    def __init__(self):
        self.chunks = []
        self.size = 0
        self.minalign = 4

    def _push(self, b):
        self.chunks.append(b)
        self.size += len(b)

    def _align(self, elem, extra=0):
        """pad so that after writing `extra` more bytes the size is a multiple of elem"""
        self.minalign = max(self.minalign, elem)
        pad = (-(self.size + extra)) % elem
        if pad:
            self._push(b'\x00' * pad)

    def byte_vector(self, data, nul=False):
        data = bytes(data)
        n = len(data)
        body = data + (b'\x00' if nul else b'')
        self._align(4, len(body))
        self._push(body)
        self._push(struct.pack('<I', n))
        return self.size

    def string(self, s):
        return self.byte_vector(s.encode(), nul=True)

    def offset_vector(self, offs):
        self._align(4, 4 * len(offs))
        # elements are written last-to-first; element i sits at end offset base - 4*i
        base = self.size + 4 * len(offs)
        out = bytearray()
        for i, o in enumerate(offs):
            pos = base - 4 * i
            out += struct.pack('<I', pos - o)
        self._push(bytes(out))
        self._push(struct.pack('<I', len(offs)))
        return self.size

    def table(self, fields):
        """fields: list of (slot, kind, value) with kind in
        'u8','u32','u64' (scalar, written even if 0 when value is not None) or 'off' (end offset)."""
        start = self.size
        placed = {}
        order = {'u64': 0, 'off': 1, 'u32': 1, 'u8': 2}
        for slot, kind, val in sorted(fields, key=lambda f: order[f[1]]):
            if val is None:
                continue
            if kind == 'u64':
                self._align(8, 8)
                self._push(struct.pack('<Q', val))
            elif kind == 'u32':
                self._align(4, 4)
                self._push(struct.pack('<I', val))
            elif kind == 'u8':
                self._push(struct.pack('<B', val))
            elif kind == 'off':
                self._align(4, 4)
                self._push(struct.pack('<I', self.size + 4 - val))
            placed[slot] = self.size
        self._align(4, 4)
        soff_chunk_index = len(self.chunks)
        self._push(b'\x00\x00\x00\x00')
        tpos = self.size
        max_slot = max(placed) if placed else 2
        n_slots = (max_slot - 4) // 2 + 1 if placed else 0
        vt = bytearray(struct.pack('<HH', 4 + 2 * n_slots, tpos - start))
        for k in range(n_slots):
            slot = 4 + 2 * k
            vt += struct.pack('<H', (tpos - placed[slot]) if slot in placed else 0)
        self._align(2, len(vt))
        self._push(bytes(vt))
        vpos = self.size
        self.chunks[soff_chunk_index] = struct.pack('<i', vpos - tpos)
        return tpos

    def finish_size_prefixed(self, root, ident=b'siev'):
        self._align(self.minalign, 12)
        self._push(ident)
        self._push(struct.pack('<I', self.size + 4 - root))
        body = b''.join(reversed(self.chunks))
        return struct.pack('<I', len(body)) + body


def _wire(b, wid):
    return b.table([(4, 'u64', wid if wid else None)])  # id 0 is the omitted default


def _value(b, data):
    return b.table([(4, 'off', b.byte_vector(data))])


def _wirelist(b, wl):
    els = []
    for e in wl:
        if isinstance(e, tuple):
            f, l = _wire(b, e[0]), _wire(b, e[1])
            r = b.table([(4, 'off', f), (6, 'off', l)])
            els.append(b.table([(4, 'u8', 2), (6, 'off', r)]))
        else:
            els.append(b.table([(4, 'u8', 1), (6, 'off', _wire(b, e))]))
    return b.table([(4, 'off', b.offset_vector(els))])


_ITER_TAG = {'const': 1, 'name': 2, 'add': 3, 'sub': 4, 'mul': 5, 'div': 6}


def _iterexpr(b, e):
    k = e[0]
    if k == 'const':
        inner = b.table([(4, 'u64', e[1] if e[1] else None)])
    elif k == 'name':
        inner = b.table([(4, 'off', b.string(e[1]))])
    elif k in ('add', 'sub', 'mul'):
        l, r = _iterexpr(b, e[1]), _iterexpr(b, e[2])
        inner = b.table([(4, 'off', l), (6, 'off', r)])
    elif k == 'div':
        n = _iterexpr(b, e[1])
        inner = b.table([(4, 'off', n), (6, 'u64', e[2] if e[2] else None)])
    else:
        raise ValueError(e)
    return b.table([(4, 'u8', _ITER_TAG[k]), (6, 'off', inner)])


def _iterlist(b, il):
    els = []
    for e in il:
        if e[0] == 'range':
            f, l = _iterexpr(b, e[1]), _iterexpr(b, e[2])
            r = b.table([(4, 'off', f), (6, 'off', l)])
            els.append(b.table([(4, 'u8', 2), (6, 'off', r)]))
        else:
            els.append(b.table([(4, 'u8', 1), (6, 'off', _iterexpr(b, e))]))
    return b.table([(4, 'off', b.offset_vector(els))])


def _u64(v):
    return v if v else None


def _gates_vector(b, gates):
    return b.offset_vector([_gate(b, g) for g in gates])


def _gate(b, g):
    k = g[0]
    if k == 'constant':
        c = b.byte_vector(g[2])
        t = b.table([(4, 'off', _wire(b, g[1])), (6, 'off', c)])
    elif k == 'assert_zero':
        t = b.table([(4, 'off', _wire(b, g[1]))])
    elif k in ('copy', 'not'):
        o, i = _wire(b, g[1]), _wire(b, g[2])
        t = b.table([(4, 'off', o), (6, 'off', i)])
    elif k in ('add', 'mul', 'and', 'xor'):
        o, l, r = _wire(b, g[1]), _wire(b, g[2]), _wire(b, g[3])
        t = b.table([(4, 'off', o), (6, 'off', l), (8, 'off', r)])
    elif k in ('addc', 'mulc'):
        c = b.byte_vector(g[3])
        o, i = _wire(b, g[1]), _wire(b, g[2])
        t = b.table([(4, 'off', o), (6, 'off', i), (8, 'off', c)])
    elif k in ('instance', 'witness'):
        t = b.table([(4, 'off', _wire(b, g[1]))])
    elif k == 'free':
        f = _wire(b, g[1])
        l = _wire(b, g[2]) if g[2] is not None else None
        t = b.table([(4, 'off', f), (6, 'off', l)])
    elif k == 'call':
        n = b.string(g[1])
        o, i = _wirelist(b, g[2]), _wirelist(b, g[3])
        t = b.table([(4, 'off', n), (6, 'off', o), (8, 'off', i)])
    elif k == 'anoncall':
        sub = _gates_vector(b, g[5])
        i = _wirelist(b, g[2])
        inner = b.table([(4, 'off', i), (6, 'u64', _u64(g[3])), (8, 'u64', _u64(g[4])), (10, 'off', sub)])
        o = _wirelist(b, g[1])
        t = b.table([(4, 'off', o), (6, 'off', inner)])
    elif k == 'switch':
        branches = []
        for br in g[4]:
            if br[0] == 'call':
                n = b.string(br[1])
                i = _wirelist(b, br[2])
                inv = b.table([(4, 'off', n), (6, 'off', i)])
                branches.append(b.table([(4, 'u8', 1), (6, 'off', inv)]))
            else:
                sub = _gates_vector(b, br[4])
                i = _wirelist(b, br[1])
                inv = b.table([(4, 'off', i), (6, 'u64', _u64(br[2])), (8, 'u64', _u64(br[3])), (10, 'off', sub)])
                branches.append(b.table([(4, 'u8', 2), (6, 'off', inv)]))
        bv = b.offset_vector(branches)
        cv = b.offset_vector([_value(b, c) for c in g[3]])
        o = _wirelist(b, g[2])
        c = _wire(b, g[1])
        t = b.table([(4, 'off', c), (6, 'off', o), (8, 'off', cv), (10, 'off', bv)])
    elif k == 'for':
        body = g[5]
        if body[0] == 'call':
            n = b.string(body[1])
            o, i = _iterlist(b, body[2]), _iterlist(b, body[3])
            bt = b.table([(4, 'off', n), (6, 'off', o), (8, 'off', i)])
            btype = 1
        else:
            sub = _gates_vector(b, body[5])
            o, i = _iterlist(b, body[1]), _iterlist(b, body[2])
            bt = b.table([(4, 'off', o), (6, 'off', i), (8, 'u64', _u64(body[3])), (10, 'u64', _u64(body[4])),
                          (12, 'off', sub)])
            btype = 2
        it = b.string(g[1])
        outs = _wirelist(b, g[4])
        t = b.table([(4, 'off', outs), (6, 'off', it), (8, 'u64', _u64(g[2])), (10, 'u64', _u64(g[3])),
                     (12, 'u8', btype), (14, 'off', bt)])
    else:
        raise ValueError(k)
    return b.table([(4, 'u8', DIRECTIVE_TAGS[k]), (6, 'off', t)])


def _header(b, modulus_bytes, degree=1, version='1.0.0'):
    fc = _value(b, modulus_bytes)
    v = b.string(version)
    return b.table([(4, 'off', v), (6, 'off', fc), (8, 'u32', degree if degree else None)])


def int_to_le(v, width=None):
    n = max(1, (v.bit_length() + 7) // 8) if width is None else width
    return v.to_bytes(n, 'little')


def write_instance(modulus, values, degree=1, version='1.0.0'):
    """values: list of bytes (little-endian Values)."""
    b = Builder()
    vec = b.offset_vector([_value(b, v) for v in values])
    h = _header(b, modulus, degree, version)
    inst = b.table([(4, 'off', h), (6, 'off', vec)])
    root = b.table([(4, 'u8', 2), (6, 'off', inst)])
    return b.finish_size_prefixed(root)


def write_witness(modulus, values, degree=1, version='1.0.0'):
    b = Builder()
    vec = b.offset_vector([_value(b, v) for v in values])
    h = _header(b, modulus, degree, version)
    wit = b.table([(4, 'off', h), (6, 'off', vec)])
    root = b.table([(4, 'u8', 3), (6, 'off', wit)])
    return b.finish_size_prefixed(root)


def write_relation(modulus, gateset, features, functions, gates, degree=1, version='1.0.0'):
    """functions: list of (name, output_count, input_count, instance_count, witness_count, [gates])."""
    b = Builder()
    gv = _gates_vector(b, gates)
    fts = []
    for (name, oc, ic, inc, wc, body) in functions:
        bv = _gates_vector(b, body)
        n = b.string(name)
        fts.append(b.table([(4, 'off', n), (6, 'u64', _u64(oc)), (8, 'u64', _u64(ic)), (10, 'u64', _u64(inc)),
                            (12, 'u64', _u64(wc)), (14, 'off', bv)]))
    fv = b.offset_vector(fts) if functions is not None else None
    feat = b.string(features)
    gs = b.string(gateset)
    h = _header(b, modulus, degree, version)
    rel = b.table([(4, 'off', h), (6, 'off', gs), (8, 'off', feat), (10, 'off', fv), (12, 'off', gv)])
    root = b.table([(4, 'u8', 1), (6, 'off', rel)])
    return b.finish_size_prefixed(root)


# ---------------------------------------------------------------------------
# Bulk writer for large "simple" relations (Add/Mul/And/Xor binary gates plus a
# python-built prologue/epilogue), vectorised with numpy: every binary gate is
# one fixed 80-byte block, so a 1M-gate message is assembled in milliseconds.
# ---------------------------------------------------------------------------
def _binary_gate_block():
    """One Directive -> Gate{output,left,right} -> 3 Wire tables, laid out as a
    self-contained block.  Returns (template bytes, hole offsets for the three
    u64 ids, offset of the Directive table inside the block)."""
    b = Builder()
    # Pad the builder so the block is 8-aligned and self-contained.
    o = b.table([(4, 'u64', 0x1111111111111111)])
    l = b.table([(4, 'u64', 0x2222222222222222)])
    r = b.table([(4, 'u64', 0x3333333333333333)])
    t = b.table([(4, 'off', o), (6, 'off', l), (8, 'off', r)])
    d = b.table([(4, 'u8', 0x7f), (6, 'off', t)])
    b._align(8, 0)
    blob = b''.join(reversed(b.chunks))
    holes = [blob.index(struct.pack('<Q', v)) for v in (0x1111111111111111, 0x2222222222222222, 0x3333333333333333)]
    assert blob.count(b'\x7f') == 1, 'tag placeholder must be unique inside the block'
    tag_off = blob.index(b'\x7f')
    d_off = len(blob) - d
    return blob, holes, tag_off, d_off


def _unary_gate_block():
    """Directive -> Gate{output,input} (GateNot / GateCopy layout) as a self-contained block."""
    b = Builder()
    o = b.table([(4, 'u64', 0x1111111111111111)])
    i = b.table([(4, 'u64', 0x2222222222222222)])
    t = b.table([(4, 'off', o), (6, 'off', i)])
    d = b.table([(4, 'u8', 0x7f), (6, 'off', t)])
    b._align(8, 0)
    blob = b''.join(reversed(b.chunks))
    holes = [blob.index(struct.pack('<Q', v)) for v in (0x1111111111111111, 0x2222222222222222)]
    assert blob.count(b'\x7f') == 1
    return blob, holes, blob.index(b'\x7f'), len(blob) - d


def write_relation_segments(modulus, gateset, features, segments, functions=None, degree=1):
    """Relation message whose directives are the concatenation of `segments`:
      ('gates', [gate tuples])                      -- any gate, built one by one
      ('bulk', tags, outs, lefts, rights)           -- numpy arrays of binary gates
                                                       (tags = DirectiveSet numbers 4 add, 5 mul, 8 and, 9 xor)
      ('bulk1', tags, outs, ins)                    -- numpy arrays of unary gates (3 copy, 10 not)
    """
    blob, holes, tag_off, d_off = _binary_gate_block()
    bs = len(blob)
    block = np.frombuffer(blob, dtype=np.uint8)
    b = Builder()
    b.minalign = 8
    parts = []  # per segment: int64 array of directive end offsets, in logical order
    for seg in segments:
        if seg[0] == 'gates':
            parts.append(np.array([_gate(b, g) for g in seg[1]], dtype=np.int64))
        elif seg[0] == 'bulk':
            _, tags, outs, lefts, rights = seg
            n = len(tags)
            arr = np.tile(block, n).reshape(n, bs)
            for h, vals in zip(holes, (outs, lefts, rights)):
                arr[:, h:h + 8] = np.ascontiguousarray(np.asarray(vals, dtype='<u8')).view(np.uint8).reshape(n, 8)
            arr[:, tag_off] = np.asarray(tags, dtype=np.uint8)
            b._align(8, 0)
            b._push(arr.tobytes())
            # block i starts at end offset size - i*bs; its Directive table sits d_off bytes in
            parts.append(b.size - np.arange(n, dtype=np.int64) * bs - d_off)
        elif seg[0] == 'bulk1':
            _, tags, outs, ins = seg
            ublob, uholes, utag, ud_off = _unary_gate_block()
            ubs = len(ublob)
            n = len(tags)
            arr = np.tile(np.frombuffer(ublob, dtype=np.uint8), n).reshape(n, ubs)
            for h, vals in zip(uholes, (outs, ins)):
                arr[:, h:h + 8] = np.ascontiguousarray(np.asarray(vals, dtype='<u8')).view(np.uint8).reshape(n, 8)
            arr[:, utag] = np.asarray(tags, dtype=np.uint8)
            b._align(8, 0)
            b._push(arr.tobytes())
            parts.append(b.size - np.arange(n, dtype=np.int64) * ubs - ud_off)
        else:
            raise ValueError(seg[0])
    offs = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
    m = len(offs)
    b._align(4, 4 * m)
    base = b.size + 4 * m
    pos = base - 4 * np.arange(m, dtype=np.int64)
    b._push((pos - offs).astype('<u4').tobytes())
    b._push(struct.pack('<I', m))
    gv = b.size
    fv = None
    if functions:
        fts = []
        for (name, oc, ic, inc, wc, body) in functions:
            bv = _gates_vector(b, body)
            nm = b.string(name)
            fts.append(b.table([(4, 'off', nm), (6, 'u64', _u64(oc)), (8, 'u64', _u64(ic)), (10, 'u64', _u64(inc)),
                                (12, 'u64', _u64(wc)), (14, 'off', bv)]))
        fv = b.offset_vector(fts)
    feat = b.string(features)
    gs = b.string(gateset)
    h = _header(b, modulus, degree)
    rel = b.table([(4, 'off', h), (6, 'off', gs), (8, 'off', feat), (10, 'off', fv), (12, 'off', gv)])
    root = b.table([(4, 'u8', 1), (6, 'off', rel)])
    return b.finish_size_prefixed(root)
