"""Test infrastructure: interprets the *scheduled device program* exported by
zkgpu_schedule_dump() (8 words per op: dst, kind + operand-expression bits, a0, a1, b0, b1, dst2, c0) with Python integers, launch by launch and slot by slot,
exactly as the HIP kernels would (Montgomery domain for odd p, bits for p=2).
It lets the CPU-only test tier check the host logic -- tape recording,
levelisation, slot reuse, constant pool -- against the oracle without a GPU.
It is not a product path and is never imported outside tests/."""

OP = {'nop': 0, 'add': 1, 'mul': 2, 'addc': 3, 'mulc': 4, 'copy': 5, 'const': 6, 'instance': 7, 'witness': 8,
      'assert': 9, 'and': 10, 'xor': 11, 'not': 12, 'nz': 13, 'carry': 14,
      'input_raw': 15, 'input_conv': 16}    # (strands: an instance / witness entry in two halves, device/args.hpp)


def is_canonical_field(p):
    """the moduli whose wires the device keeps as canonical residues (the any-modulus kernels) unless told otherwise:
    even ones other than 2 and those wider than 512 bits (csrc/tape.cpp FieldHost::init)"""
    return (p % 2 == 0 and p != 2) or p.bit_length() > 512


def simulate(ops, launches, const_words, words_per_const, n_slots, p, instances, witnesses, shuffle_seed=None, modes=None,
             carries=(), canonical=None, n_raw_consts=0):
    """Run one lane.  instances / witnesses: python ints.  Returns (slots, first_fail_seq, noncanonical).
    modes = (instance modes, witness modes) as zkgpu_input_modes gives them (how a value >= p is treated per position:
    0xFF flags the lane; GF(2): 0x01 packs `v != 0`); None = every position 0.
    Within a non-sequential launch the ops are executed in a shuffled order when shuffle_seed is given
    (they must be independent), mimicking the arbitrary order of waves on the GPU."""
    import random
    if canonical is None:
        canonical = is_canonical_field(p)
    boolean = p == 2 and not canonical
    if canonical:   # zkgpu_field_representation 2: no Montgomery factor; a value must still fit the limbs
        R, rinv, wide = 1, 1, 1 << (32 * words_per_const)
    elif not boolean:
        nbits = 32 * words_per_const
        R = 1 << nbits
        rinv = pow(R, -1, p)
        wide = R
    consts = []
    for i in range(len(const_words) // max(words_per_const, 1)):
        v = 0
        for k in range(words_per_const):
            v |= int(const_words[i * words_per_const + k]) << (32 * k)
        consts.append(v)
    mode_lists = list(modes) if modes is not None else []
    while len(mode_lists) < 3:
        mode_lists.append([])
    # stream 3 of the source codes: the last n_raw_consts entries of the pool are constants >= p kept as plain integers
    streams = (instances, witnesses, carries, consts[len(consts) - n_raw_consts:] if n_raw_consts else [])

    def mode_of(stream, position):
        m = mode_lists[stream]
        return m[position] if position < len(m) else 0

    def source_is_nonzero(code):
        """the unreduced source an assert_zero / not entry names (device/replay_kernels.hpp unreduced_source_is_nonzero)"""
        if code < 2:
            return code == 1
        q = code - 2
        v = streams[q & 3][q >> 2]
        return v >= p
    slots = _Slots(n_slots)
    first_fail = None
    noncanon = False
    rng = random.Random(shuffle_seed)
    for (first, count, opw, sequential) in launches:
        slots.new_launch(bool(sequential))
        idx = list(range(int(first), int(first) + int(count)))
        if shuffle_seed is not None and not sequential:
            rng.shuffle(idx)
        pending = []
        reads = set()
        for i in idx:
            dst, kbits, a, a1, b, b1, dst2, c0 = (int(x) for x in ops[i][:8])
            kind, ea, eb, second = kbits & 0xFF, (kbits >> 8) & 3, (kbits >> 10) & 3, (kbits >> 12) & 3

            def operand(x0, x1, e):  # a slot, or add / mul of two slots evaluated "in registers" (gate fusion)
                reads.add(x0)
                v0 = slots[x0]
                assert v0 is not None, 'read of an unwritten slot'
                if not e:
                    return v0
                reads.add(x1)
                v1 = slots[x1]
                assert v1 is not None, 'read of an unwritten slot'
                return (v0 + v1) % p if e == 1 else v0 * v1 * rinv % p
            if kind in (OP['and'], OP['xor']) and not boolean:
                # integer bit operation, then % p (evaluator.rs:924-933).  An operand is the canonical value of a wire or,
                # reference 0x80000000 | code, the RAW value of the input it is a copy of (device/args.hpp kOperandIsSource)
                vals = []
                for ref in (a, b):
                    if ref & 0x80000000:
                        q = (ref & 0x7FFFFFFF) - 2
                        v = streams[q & 3][q >> 2]
                        if v >= wide:
                            noncanon = True
                        vals.append(v % wide)
                    else:
                        reads.add(ref)
                        assert slots[ref] is not None, 'read of an unwritten slot'
                        vals.append(slots[ref] * rinv % p)
                r = ((vals[0] & vals[1]) if kind == OP['and'] else (vals[0] ^ vals[1])) % p * R % p
            elif kind in (OP['add'], OP['mul'], OP['and'], OP['xor']):
                x, y = operand(a, a1, ea), operand(b, b1, eb)
                if kind == OP['add']:
                    r = (x + y) % p
                elif kind == OP['mul']:
                    r = x * y * rinv % p
                else:
                    r = x & y if kind == OP['and'] else x ^ y
                if second:  # pair entry: a second gate of the level shares operand a
                    assert kind in (OP['add'], OP['mul'])
                    reads.add(c0)
                    z = slots[c0]
                    assert z is not None, 'read of an unwritten slot'
                    r2 = (x + z) % p if second == 1 else x * z * rinv % p
                    if sequential:
                        assert dst2 != dst
                        slots[dst2] = r2
                    else:
                        pending.append((dst2, r2))
            elif kind in (OP['addc'], OP['mulc']):
                x = slots[a]
                reads.add(a)
                assert x is not None
                r = (x + consts[b]) % p if kind == OP['addc'] else x * consts[b] * rinv % p
            elif kind == OP['copy']:
                r = slots[a]
                reads.add(a)
                assert r is not None
            elif kind == OP['nz']:   # x^(p-1) of a prime field: 1 (device form) unless x is 0
                assert slots[a] is not None and not boolean
                reads.add(a)
                r = 0 if slots[a] == 0 else R % p
            elif kind == OP['not']:
                assert slots[a] is not None
                reads.add(a)
                code = 0 if boolean else (a1 | b)    # fused entry: a1; unfused: b (the other is 0)
                r = 1 - slots[a] if boolean else (R % p if slots[a] == 0 and not source_is_nonzero(code) else 0)
            elif kind == OP['const']:
                r = consts[a]
            elif kind == OP['input_raw']:     # the words of the input as they lie in the buffer, into an LDS value
                assert sequential and dst & K_SLOT_IN_LDS and a1 in (0, 1)
                r = ('raw', a1, a)
            elif kind in (OP['instance'], OP['witness'], OP['carry'], OP['input_conv']):
                if kind == OP['input_conv']:  # ... and their conversion: position b of stream a1, read back from LDS value a
                    assert slots[a] == ('raw', a1, b), 'the conversion reads an LDS value its fetch did not write'
                    reads.add(a)
                    stream, a = a1, b
                else:
                    stream = {OP['instance']: 0, OP['witness']: 1, OP['carry']: 2}[kind]
                v = streams[stream][a]
                mode = mode_of(stream, a)
                # wider than the limbs of this field: reduced like any other value where only arithmetic reads it (the
                # reference's gates are `% m` of whatever integer comes in); flagged where its bits matter
                if not boolean and v >= wide and mode in (0xFF, 0x03):
                    noncanon = True
                if v >= p and mode == 0xFF:     # the unreduced value would reach an integer bit operation / Evaluator::get
                    noncanon = True
                r = ((1 if v else 0) if mode == 0x01 else (v & 1)) if boolean else (v * R % p)
            elif kind == OP['assert']:
                assert slots[a] is not None
                reads.add(a)
                code = 0 if boolean else (dst | a1)   # fused entry: a1; unfused: dst (the other is 0)
                if (slots[a] != 0 or source_is_nonzero(code)) and (first_fail is None or b < first_fail):
                    first_fail = b
                continue
            else:
                continue
            if sequential:
                slots[dst] = r
            else:
                pending.append((dst, r))  # a level's writes never feed its own reads
        written = [d for d, _ in pending]
        assert len(set(written)) == len(written), 'two ops of one level write the same slot'
        assert not (reads & set(written)), 'a level overwrites a slot it also reads (race on the GPU)'
        for dst, r in pending:
            slots[dst] = r
    return slots, first_fail, noncanon


K_SLOT_IN_LDS = 0x40000000


class _Slots:
    """the wire table + the LDS of the workgroup that walks a strand: a slot number with K_SLOT_IN_LDS names a value that
    lives in LDS for the duration of ONE sequential launch (csrc/schedule.hpp kSlotInLds); anywhere else it is an error"""

    def __init__(self, n):
        self.table = [None] * n
        self.lds = {}
        self.in_strand = False

    def new_launch(self, sequential):
        self.lds = {}
        self.in_strand = sequential

    def __getitem__(self, k):
        if k & K_SLOT_IN_LDS:
            assert self.in_strand, 'an LDS slot outside a strand'
            return self.lds.get(k & ~K_SLOT_IN_LDS)
        return self.table[k]

    def __setitem__(self, k, v):
        if k & K_SLOT_IN_LDS:
            assert self.in_strand, 'an LDS slot outside a strand'
            self.lds[k & ~K_SLOT_IN_LDS] = v
        else:
            self.table[k] = v

    def __len__(self):
        return len(self.table)

    def __iter__(self):
        return iter(self.table)


def from_device_form(v, p, words_per_const, canonical=None):
    if canonical is None:
        canonical = is_canonical_field(p)
    if (p == 2 and not canonical) or v is None or canonical:
        return v
    return v * pow(1 << (32 * words_per_const), -1, p) % p


# ---- strands: a static check of what the kernel's execution model needs ------------------------------------------------
def strand_hazards(ops, first, level_ptr):
    """A strand (replay_strand_kernel) runs the entries of a level on four waves at once -- entry i of the level on wave
    i % 4, a wave its own entries in order -- with a barrier between levels.  `simulate` runs them one after the other, so
    its result is the kernel's only if no entry of a level touches what an entry of ANOTHER wave of that level writes.
    Returns the list of violations (empty: the sequential run is what the GPU computes)."""
    K = OP
    one_operand = (K['addc'], K['mulc'], K['copy'], K['nz'], K['not'], K['assert'], K['input_conv'])

    def reads(o):
        kind, ea, eb, pair = int(o[1]) & 0xFF, (int(o[1]) >> 8) & 3, (int(o[1]) >> 10) & 3, (int(o[1]) >> 12) & 3
        if kind in (K['add'], K['mul']):
            r = [int(o[2]), int(o[4])]
            if ea:
                r.append(int(o[3]))
            if eb:
                r.append(int(o[5]))
            if pair:
                r.append(int(o[7]))
            return r
        if kind in one_operand:
            return [int(o[2])]
        if kind in (K['and'], K['xor']):
            return [int(x) for x in (o[2], o[4]) if not int(x) & 0x80000000]
        return []

    def writes(o):
        kind, pair = int(o[1]) & 0xFF, (int(o[1]) >> 12) & 3
        if kind in (K['assert'], 0):
            return []
        return [int(o[0])] + ([int(o[6])] if pair and kind in (K['add'], K['mul']) else [])

    bad = []
    for l in range(len(level_ptr) - 1):
        b, e = first + int(level_ptr[l]), first + int(level_ptr[l + 1])
        writer = {}
        for i in range(b, e):
            for w in writes(ops[i]):
                if w in writer and writer[w] != (i - b) % 4:
                    bad.append('level %d: entries of waves %d and %d both write slot %#x' % (l, writer[w], (i - b) % 4, w))
                writer[w] = (i - b) % 4
        for i in range(b, e):
            for r in reads(ops[i]):
                if r in writer and writer[r] != (i - b) % 4:
                    bad.append('level %d: wave %d reads slot %#x that wave %d writes' % (l, (i - b) % 4, r, writer[r]))
    return bad
