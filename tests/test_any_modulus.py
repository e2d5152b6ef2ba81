"""The any-modulus path (csrc/device/generic_kernels.hpp): PlaintextBackend takes every BigUint modulus
(rust/src/consumers/evaluator.rs:866-938), the Montgomery kernels only odd ones of at most 512 bits.  Even
characteristics other than 2, characteristics of up to 4096 bits, and GF(2) in a session that also works in another field
(evaluator.rs:232-237) run on kernels that keep canonical residues and reduce with Barrett's method.

CPU tier: the kernels' arithmetic itself, run on the host through zkgpu_generic_selftest, against Python integers; random
structured relations through recording, scheduling and the program interpreter against the oracle; sessions that change
between GF(2) and another field against the oracle.  GPU tier: the same relations on the card, every lane against the
oracle."""
import random

import pytest

import program_sim
from helpers import batch_arrays, oracle_lane
from oracle_lib import OracleRun
from random_circuits import Gen
from test_fuzz_host import expected_product_violations
import zkinterface_ir_amd as zk
from zkinterface_ir_amd import sieve_writer as sw


def _odd(bits, seed):
    return random.Random(seed).getrandbits(bits) | (1 << (bits - 1)) | 1


# (modulus, Boolean gate set): even, powers of 2^32, wider than 512 bits (prime and composite), the width limit
MODULI = [(6, False), (2 ** 32, False), (2 ** 64, True), (2 ** 64 - 2, False), (2 * 101 * (2 ** 61 - 1), False), (2 ** 521 - 1, False),
          (_odd(600, 1), True), (_odd(1024, 2) - 1, False), (2 ** 1279 - 1, False), (_odd(2048, 3), False), (2 ** 4096 - 1, False),
          (4, True), (2 ** 96, False), (_odd(513, 4), False)]


def _width(p):
    return 8 * ((p.bit_length() + 63) // 64)


def test_arithmetic_of_the_kernels_against_python_integers():
    rnd = random.Random(11)
    mods = [2, 3, 4, 6, 2 ** 31, 2 ** 32, 2 ** 32 - 1, 2 ** 32 + 1, 2 ** 64, 2 ** 64 - 2, 101, 2 ** 255 - 19, 2 ** 521 - 1, 2 ** 96,
            2 ** 128, 2 ** 4095, 2 ** 4096 - 1,
            # powers of two: the rings Z / 2^B take the low B bits instead of a Barrett reduction (generic_kernels.hpp g_low_bits)
            8, 2 ** 33, 2 ** 63, 2 ** 65, 2 ** 100, 2 ** 127, 2 ** 256, 2 ** 1000]
    for bits in (2, 3, 17, 31, 32, 33, 63, 64, 65, 95, 96, 97, 128, 255, 256, 511, 512, 513, 600, 1024, 2047, 2048, 4096):
        mods += [rnd.getrandbits(bits) | (1 << (bits - 1)) | 1, (rnd.getrandbits(bits) | (1 << (bits - 1))) & ~1]
    for p in mods:
        if p < 2:
            continue
        nw = 2 * ((p.bit_length() + 63) // 64)
        for it in range(5):
            a, b = (p - 1, p - 1) if it == 0 else (0, rnd.randrange(p)) if it == 1 else (rnd.randrange(p), rnd.randrange(p))
            assert zk.generic_selftest(p, 'add', a, b) == (a + b) % p, ('add', p, a, b)
            assert zk.generic_selftest(p, 'mul', a, b) == (a * b) % p, ('mul', p, a, b)
            assert zk.generic_selftest(p, 'and', a, b) == (a & b) % p, ('and', p, a, b)
            assert zk.generic_selftest(p, 'xor', a, b) == (a ^ b) % p, ('xor', p, a, b)
            raw = (1 << (32 * nw)) - 1 if it == 0 else rnd.getrandbits(32 * nw)
            assert zk.generic_selftest(p, 'reduce', raw) == raw % p, ('reduce', p, raw)
    for bad in (0, 1, 2 ** 4096 + 1):
        with pytest.raises(zk.ZkGpuError):
            zk.generic_selftest(bad, 'add', 0, 0)


def test_which_fields_take_which_path():
    for p, want in ((2, 0), (101, 1), (2 ** 512 - 569, 1), (6, 2), (2 ** 64, 2), (2 ** 521 - 1, 2), (2 ** 4096 - 1, 2)):
        ev = zk.Evaluator()
        ev.declare_inputs(0, 1)
        ev.ingest_message(sw.write_relation(sw.int_to_le(p), 'arithmetic', 'simple', [], [('witness', 0), ('assert_zero', 0)]))
        assert ev.host_violations() == []
        assert ev.field_representation(0) == want, p
        assert ev.elem_bytes == (1 if p == 2 else _width(p))
    for p, text in ((2 ** 4096 + 1, 'wider than 4096 bits'), (1, 'characteristic 1')):
        ev = zk.Evaluator()
        ev.ingest_message(sw.write_relation(sw.int_to_le(p), 'arithmetic', 'simple', [], [('witness', 0), ('assert_zero', 0)]))
        assert any(text in m for m in ev.host_violations()), ev.host_violations()


@pytest.mark.parametrize('seed', range(28))
def test_random_relation_against_oracle(seed):
    """the host side (recording, scheduling: unfused entries, canonical constants) under the program interpreter"""
    p, boolean = MODULI[seed % len(MODULI)]
    g = Gen(seed + 500, p, boolean)
    rel, mod_le = g.relation()
    rows_i, rows_w = g.lane_inputs(3, seed + 1000)
    w = _width(p)
    for retain in (True, False):
        ev = zk.Evaluator()
        ev.declare_inputs(g.n_inst, g.n_wit)
        ev.ingest_message(rel)
        if not ev.n_value_ops and ev.host_violations():
            assert ev.host_violations() == oracle_lane(mod_le, rows_i[0], rows_w[0], [rel], w).violations
            return
        assert ev.field_representation(0) == 2
        ev.finalize(retain_all=retain)
        ops, launches, consts, slot_of = ev.schedule_dump()
        info = ev.schedule_info()
        assert info['words_per_const'] == w // 4
        kinds, _, _ = ev.tape()
        for lane in range(3):
            ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], w)
            slots, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p, rows_i[lane],
                                                       rows_w[lane], shuffle_seed=seed, modes=(ev.input_modes(False), ev.input_modes(True)))
            assert not noncanon
            assert expected_product_violations(ev, ff) == ref.violations, (seed, lane)
            if retain:
                vals = [slots[slot_of[i]] for i in range(len(kinds)) if kinds[i] != 9]
                rv = ref.trace_values()
                assert vals[:len(rv)] == rv, (seed, lane)


# ---- GF(2) beside another field ----------------------------------------------------------------------------------------
P = 2 ** 61 - 1


def _mixed_statement(order):
    """Two or three Relation messages, in GF(2) and in GF(P); wires cross every boundary.  Returns (messages, function
    lane -> ([instances], [witnesses], satisfied))."""
    two, big = sw.int_to_le(2), sw.int_to_le(P)
    if order == 'bool_first':
        msgs = [sw.write_relation(two, 'boolean', 'simple', [],
                                  [('witness', 0), ('witness', 1), ('and', 2, 0, 1), ('xor', 3, 0, 1), ('not', 4, 3), ('free', 0, 1)]),
                # 2, 3, 4 arrive as the integers 0 / 1
                sw.write_relation(big, 'arithmetic', 'simple', [],
                                  [('witness', 5), ('mul', 6, 5, 5), ('add', 7, 6, 2), ('add', 8, 7, 3), ('add', 9, 8, 4), ('instance', 10),
                                   ('mulc', 11, 10, sw.int_to_le(P - 1)), ('add', 12, 9, 11), ('assert_zero', 12), ('free', 2, 12)])]

        def lane(k):
            a, b, x = k & 1, (k >> 1) & 1, 1000 + 77 * k
            e = (x * x + (a & b) + (a ^ b) + (1 - (a ^ b))) % P
            ok = k % 3 != 2
            return [e if ok else (e + 1) % P], [a, b, x], ok
        return msgs, lane
    if order == 'bool_last':
        msgs = [sw.write_relation(big, 'arithmetic', 'simple', [],
                                  [('witness', 0), ('witness', 1), ('mul', 2, 0, 1), ('addc', 3, 2, sw.int_to_le(5)), ('free', 0, 2)]),
                # 3 arrives as an integer far above 1: over GF(2) `xor` takes its low bit, `not` asks whether it is zero
                sw.write_relation(two, 'boolean', 'simple', [],
                                  [('witness', 4), ('xor', 5, 3, 4), ('not', 6, 3), ('instance', 7), ('xor', 8, 5, 7), ('assert_zero', 8),
                                   ('assert_zero', 6), ('free', 3, 8)])]

        def lane(k):
            x, y, t = 3 + k, 11 + 5 * k, k & 1
            v = (x * y + 5) % P
            bit = (v ^ t) & 1
            ok = k % 4 != 1
            return [bit if ok else 1 - bit], [x, y, t], ok   # (`not 3` is 0 for every lane: v is never 0)
        return msgs, lane
    assert order == 'there_and_back'
    msgs = [sw.write_relation(two, 'boolean', 'simple', [], [('witness', 0), ('witness', 1), ('xor', 2, 0, 1), ('free', 0, 1)]),
            sw.write_relation(big, 'arithmetic', 'simple', [], [('witness', 3), ('add', 4, 3, 2), ('mul', 5, 4, 4), ('free', 2, 4)]),
            sw.write_relation(two, 'boolean', 'simple', [], [('instance', 6), ('xor', 7, 5, 6), ('assert_zero', 7), ('free', 5, 7)])]

    def lane(k):
        a, b, x = k & 1, (k >> 2) & 1, 9 + k
        bit = (((x + (a ^ b)) ** 2) % P) & 1
        ok = k % 5 != 0
        return [bit if ok else 1 - bit], [a, b, x], ok
    return msgs, lane


def _statement(msgs, inst, wit, width=8):
    return [sw.write_instance(sw.int_to_le(2), [sw.int_to_le(v, width) for v in inst]),
            sw.write_witness(sw.int_to_le(2), [sw.int_to_le(v, width) for v in wit])] + msgs


@pytest.mark.parametrize('order', ['bool_first', 'bool_last', 'there_and_back'])
def test_session_between_gf2_and_another_field_host_side(order):
    msgs, lane = _mixed_statement(order)
    ev = zk.Evaluator()
    ev.declare_inputs(1, 3)
    for m in msgs:
        ev.ingest_message(m)
    assert ev.host_violations() == []
    n_seg = ev.n_field_segments
    assert n_seg == len(msgs)
    moduli = [2, P, 2][:n_seg] if order != 'bool_last' else [P, 2]
    # GF(2) is kept as integers in such a session; the odd field stays on the Montgomery kernels
    assert [ev.field_representation(k) for k in range(n_seg)] == [2 if m == 2 else 1 for m in moduli]
    assert ev.elem_bytes == 8
    ev.finalize()
    for k in range(7):
        inst, wit, ok = lane(k)
        ref = OracleRun(buffers=_statement(msgs, inst, wit))
        assert (ref.violations == []) == ok, (order, k, ref.violations)
        from test_field_segments import _simulate
        ff, flagged = _simulate(ev, moduli, inst, wit)
        assert not flagged
        assert (ff is None) == ok, (order, k)


# ---- GPU tier ----------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(len(MODULI)))
def test_random_relations_on_gpu(seed):
    """every lane's wire values and violation strings (retain_all), then the production schedule's verdicts"""
    p, boolean = MODULI[seed]
    # (a Switch case is a ladder of bits(p) dependent products: left to the moduli where that is a few thousand small ones)
    g = Gen(seed + 500, p, boolean, switches=p.bit_length() <= 700)
    rel, mod_le = g.relation(n_top=12)
    lanes = 67
    rows_i, rows_w = g.lane_inputs(lanes, seed + 1000)
    w = _width(p)
    refs = [oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], w) for lane in range(lanes)]
    for retain in (True, False):
        ev = zk.Evaluator()
        ev.declare_inputs(g.n_inst, g.n_wit)
        ev.ingest_message(rel)
        ev.finalize(retain_all=retain)
        assert ev.elem_bytes == w
        inst, wit = batch_arrays(rows_i, rows_w, w)
        ev.set_inputs(inst if g.n_inst else None, wit if g.n_wit else None, lanes)
        ev.replay()
        ev.synchronize()
        vals = ev.dump_trace_values(lanes) if retain else None
        n_ok = 0
        for lane in range(lanes):
            assert ev.get_violations(lane) == refs[lane].violations, (seed, lane)
            if retain:
                rv = refs[lane].trace_values()
                assert vals[lane][:len(rv)] == rv, (seed, lane)
            n_ok += not refs[lane].violations
        assert ev.counts() == (n_ok, lanes - n_ok)


@pytest.mark.gpu
def test_values_at_and_above_the_modulus_on_gpu():
    """inputs >= p over an even modulus: reduced where they meet arithmetic, non-zero for assert_zero / not, the lane
    flagged where an integer bit operation would need the unreduced bits (tests/test_unreduced.py for the odd fields)"""
    p = 2 ** 64 - 2
    rel = sw.write_relation(sw.int_to_le(p), 'arithmetic', 'simple', [],
                            [('witness', 0), ('witness', 1), ('add', 2, 0, 1), ('instance', 3), ('mulc', 4, 3, sw.int_to_le(p - 1)),
                             ('add', 5, 2, 4), ('assert_zero', 5), ('witness', 6), ('copy', 7, 6), ('assert_zero', 7), ('free', 0, 7)])
    rows_w = [[5, 7, 0], [p + 1, 7, 0], [5, 7, p], [2 ** 64 - 1, 2 ** 64 - 1, 0]]
    rows_i = [[12], [8], [12], [2]]
    ev = zk.Evaluator()
    ev.declare_inputs(1, 3)
    ev.ingest_message(rel)
    ev.finalize()
    inst, wit = batch_arrays(rows_i, rows_w, 8)
    ev.set_inputs(inst, wit, 4)
    ev.replay()
    ev.synchronize()
    for lane in range(4):
        ref = oracle_lane(sw.int_to_le(p), rows_i[lane], rows_w[lane], [rel], 8)
        assert ev.get_violations(lane) == ref.violations, lane
    assert ev.counts() == (3, 1)   # lane 2: the witness p is not zero as an integer


@pytest.mark.gpu
@pytest.mark.parametrize('order', ['bool_first', 'bool_last', 'there_and_back'])
def test_session_between_gf2_and_another_field_on_gpu(order):
    msgs, lane = _mixed_statement(order)
    lanes = 70
    rows = [lane(k) for k in range(lanes)]
    ev = zk.Evaluator()
    ev.declare_inputs(1, 3)
    for m in msgs:
        ev.ingest_message(m)
    ev.finalize()
    inst, wit = batch_arrays([r[0] for r in rows], [r[1] for r in rows], ev.elem_bytes)
    ev.set_inputs(inst, wit, lanes)
    ev.replay()
    ev.synchronize()
    n_ok = 0
    for k in range(lanes):
        ref = OracleRun(buffers=_statement(msgs, rows[k][0], rows[k][1]))
        assert (ref.violations == []) == rows[k][2], (order, k)
        assert ev.get_violations(k) == ref.violations, (order, k)
        n_ok += rows[k][2]
    assert ev.counts() == (n_ok, lanes - n_ok)


@pytest.mark.gpu
# (three words on the compile-time instantiation, 19 on the general kernel; the rings Z / 2^64 and Z / 2^128 and an even
# characteristic of eight words: replay_generic_kernel<8, 2 | 3 | 4 | 5 | 8>)
@pytest.mark.parametrize('p', [2 ** 89 - 2, 2 ** 607 - 1, 2 ** 64, 2 ** 128, 2 ** 256 - 2, 2 ** 33])
def test_wide_levels_and_a_full_batch_on_gpu(p):
    """1024 lanes (16 lane blocks: the XCD-aware grid, two streams) over levels of 48 gates, the verdict of every lane against
    Python integers; the product of the last layer is pinned by an instance value per lane"""
    rnd = random.Random(p % 1009)
    W, D, lanes = 48, 3, 1024
    gates = [('witness', k) for k in range(W)]
    nid = W
    prev = list(range(W))
    layers = []
    for _ in range(D):
        cur, spec = [], []
        for _ in range(W):
            op, x, y = rnd.choice(['add', 'mul']), rnd.choice(prev), rnd.choice(prev)
            gates.append((op, nid, x, y))
            spec.append((op, prev.index(x), prev.index(y)))
            cur.append(nid)
            nid += 1
        layers.append(spec)
        prev = cur
    acc = prev[0]
    for w in prev[1:]:
        gates.append(('add', nid, acc, w))
        acc = nid
        nid += 1
    gates += [('instance', nid), ('mulc', nid + 1, nid, sw.int_to_le(p - 1)), ('add', nid + 2, acc, nid + 1), ('assert_zero', nid + 2),
              ('free', 0, nid + 2)]
    rel = sw.write_relation(sw.int_to_le(p), 'arithmetic', 'simple', [], gates)
    rows_w = [[rnd.randrange(p) for _ in range(W)] for _ in range(lanes)]
    rows_i = []
    for lane, row in enumerate(rows_w):
        vals = row
        for spec in layers:
            vals = [(vals[x] + vals[y]) % p if op == 'add' else (vals[x] * vals[y]) % p for op, x, y in spec]
        rows_i.append([(sum(vals) + (1 if lane % 37 == 5 else 0)) % p])
    ev = zk.Evaluator()
    ev.declare_inputs(1, W)
    ev.ingest_message(rel)
    assert ev.host_violations() == [] and ev.field_representation(0) == 2
    ev.finalize()
    inst, wit = batch_arrays(rows_i, rows_w, ev.elem_bytes)
    ev.set_inputs(inst, wit, lanes)
    ev.replay()
    ev.synchronize()
    first, flags = ev.lane_results(lanes)
    bad = [lane for lane in range(lanes) if lane % 37 == 5]
    assert [lane for lane in range(lanes) if int(first[lane]) != zk.NO_FAIL] == bad and not flags.any()
    assert ev.counts() == (lanes - len(bad), len(bad))
    ev.set_lane_group(512)      # lane groups one after the other: the same verdicts
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (lanes - len(bad), len(bad))
