"""Values >= the field characteristic.  PlaintextBackend keeps constants, instance and witness
values unreduced (rust/src/consumers/evaluator.rs:862-864,896-898,940-946): add / mul / add_constant / mul_constant
reduce their result, but `copy` clones the integer as it is, `assert_zero` and `not` test it for zero, `and` / `xor`
work on its bits and `Evaluator::get` returns it.  The product gives such a value the reference's verdict wherever that
is decidable without the integer itself: it reduces it exactly where the reference's own arithmetic would (anything
that first meets an arithmetic gate; over GF(2) also and / xor, whose low bit only depends on the low bits), and an
assert_zero / not reached through copies alone sees "not zero" -- a value >= p is never the integer 0 -- by testing the
raw input beside the wire (GF(p)) or by packing the position as `v != 0` (GF(2)).  What is left is refused, and the
refusal says so: the bits of an unreduced integer in and / xor over an odd field, Evaluator::get of a wire alive at
the end, and a GF(2) position one bit cannot serve (zero test and gate at once) -- per lane for inputs, at finalize for
constants.  Every case below is held against the oracle's run of the same statement."""
import pytest

import circuits
import program_sim
from helpers import oracle_lane
from test_fuzz_host import expected_product_violations
import zkinterface_ir_amd as zk
from zkinterface_ir_amd import sieve_writer as sw

P = 101
BIG = circuits.BN254_R


def _run(p, gates, inst, wit, gateset='arithmetic', width=None):
    """(product verdict through program_sim, oracle violations, flagged-as-non-canonical)"""
    mod_le = sw.int_to_le(p)
    width = width or max(4, 8 * ((p.bit_length() + 63) // 64))
    rel = sw.write_relation(mod_le, gateset, 'simple', [], gates)
    ev = zk.Evaluator()
    ev.declare_inputs(len(inst), len(wit))
    ev.ingest_message(rel)
    ev.finalize()
    ops, launches, consts, _ = ev.schedule_dump()
    info = ev.schedule_info()
    modes = (ev.input_modes(False), ev.input_modes(True))
    n_raw = len(consts) // max(info['words_per_const'], 1) - ev.n_constants if p != 2 else 0
    _, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p, inst, wit, modes=modes,
                                           n_raw_consts=n_raw)
    ref = oracle_lane(mod_le, inst, wit, [rel], max(width, 32), trace=False)
    _run.modes = modes
    return expected_product_violations(ev, ff), ref.violations, noncanon


@pytest.mark.parametrize('p', [P, BIG])
def test_inputs_that_first_meet_arithmetic_are_reduced_like_the_reference_does(p):
    # w * w - e == 0 with w = p + 3 (>= p) and e = 9 + 2p: every use is an arithmetic gate
    # (the wires are freed: a wire alive at the end can be asked for with Evaluator::get, unreduced, and stays strict)
    gates = [('witness', 0), ('instance', 1), ('mul', 2, 0, 0), ('mulc', 3, 1, sw.int_to_le(p - 1)), ('add', 4, 2, 3),
             ('assert_zero', 4), ('free', 0, 4)]
    for w, e, true in ((p + 3, 9 + 2 * p, True), (3, 9, True), (p + 3, 10, False), (2 * p + 4, 16 + p, True)):
        if max(w, e).bit_length() > 8 * ((p.bit_length() + 63) // 64) * 1:   # must fit the input width
            continue
        mine, ref, noncanon = _run(p, gates, [e], [w])
        assert not noncanon
        assert mine == ref and (ref == []) == true, (w, e)


def test_an_unreduced_input_that_reaches_assert_zero_through_copies_is_not_zero():
    # assert_zero(copy(copy(w))): the reference tests the unreduced integer: w = p, 2p are NOT zero for it
    gates = [('witness', 0), ('copy', 1, 0), ('copy', 2, 1), ('assert_zero', 2), ('free', 0, 2)]
    for w, true in ((0, True), (P, False), (2 * P, False), (5, False), (P + 5, False)):
        mine, ref, noncanon = _run(P, gates, [], [w])
        assert not noncanon and mine == ref, w
        assert ref == ([] if true else ['Wire_2 (may be weighted) should be 0, while it is not'])
        assert _run.modes == ([], [0x01])                           # read by zero tests alone
    # the same input through an arithmetic gate first: p + 0 = 0, both say TRUE
    gates = [('witness', 0), ('addc', 1, 0, bytes([0])), ('assert_zero', 1), ('free', 0, 1)]
    mine, ref, noncanon = _run(P, gates, [], [P])
    assert mine == ref == [] and not noncanon and _run.modes == ([], [0x00])
    # zero test AND arithmetic on one witness: the sink sees the integer, the product sees the residue
    gates = [('witness', 0), ('instance', 1), ('copy', 2, 0), ('assert_zero', 2), ('mul', 3, 0, 0), ('mulc', 4, 1, sw.int_to_le(P - 1)),
             ('add', 5, 3, 4), ('assert_zero', 5), ('free', 0, 5)]
    for w, e in ((0, 0), (P, 0), (P, 1), (P + 3, 9), (3, 9)):
        mine, ref, noncanon = _run(P, gates, [e], [w])
        assert not noncanon and mine == ref, (w, e)
        assert _run.modes == ([0x00], [0x02])
    assert _run(P, gates, [0], [P])[1] == ['Wire_2 (may be weighted) should be 0, while it is not']
    # over BN254: the raw value is compared with a 254-bit p word by word
    gates = [('instance', 0), ('copy', 1, 0), ('assert_zero', 1), ('free', 0, 1)]
    for v in (0, BIG, BIG + 1, 2 ** 256 - 1, 1):
        mine, ref, noncanon = _run(BIG, gates, [v], [])
        assert not noncanon and mine == ref and (ref == []) == (v == 0), v


def test_not_sees_the_unreduced_integer():
    # not = is_zero ? 1 : 0 on the integer (evaluator.rs:935-938): not(p) = 0, not(0) = 1
    gates = [('witness', 0), ('not', 1, 0), ('assert_zero', 1), ('free', 0, 1)]
    for w in (0, 1, P, 2 * P, P + 1):
        mine, ref, noncanon = _run(P, gates, [], [w], gateset='arithmetic,boolean')
        assert not noncanon and mine == ref and (ref == []) == (w != 0), w
    # through a copy, and the result used by arithmetic
    gates = [('witness', 0), ('copy', 1, 0), ('not', 2, 1), ('mulc', 3, 2, bytes([7])), ('addc', 4, 3, sw.int_to_le(P - 7)), ('assert_zero', 4),
             ('free', 0, 4)]
    for w in (0, P, 3):
        mine, ref, noncanon = _run(P, gates, [], [w], gateset='arithmetic,boolean')
        assert not noncanon and mine == ref and (ref == []) == (w == 0), w


def test_bit_operations_see_the_unreduced_integer_over_an_odd_field_and_the_low_bit_over_gf2():
    # odd p: (w & 3) on the unreduced integer -- the entry reads the raw input, not the wire (mode 0x03)
    gates = [('witness', 0), ('constant', 1, bytes([3])), ('and', 2, 0, 1), ('addc', 3, 2, sw.int_to_le(P - 1)), ('assert_zero', 3)]
    mine, ref, noncanon = _run(P, gates, [], [5], gateset='arithmetic,boolean')      # 5 & 3 = 1
    assert mine == ref == [] and not noncanon and _run.modes == ([], [0x03])
    mine, ref, noncanon = _run(P, gates, [], [P + 5], gateset='arithmetic,boolean')      # (106 & 3) = 2 for the reference
    assert mine == ref != [] and not noncanon
    # through copies, both operands inputs, the result beyond p before `% p`; the same input also in arithmetic
    gates = [('witness', 0), ('instance', 1), ('copy', 2, 0), ('copy', 3, 2), ('xor', 4, 3, 1), ('and', 5, 1, 2), ('mul', 6, 0, 0),
             ('witness', 7), ('witness', 8), ('witness', 9),
             ('mulc', 10, 7, sw.int_to_le(P - 1)), ('add', 11, 4, 10), ('assert_zero', 11),
             ('mulc', 12, 8, sw.int_to_le(P - 1)), ('add', 13, 5, 12), ('assert_zero', 13),
             ('mulc', 14, 9, sw.int_to_le(P - 1)), ('add', 15, 6, 14), ('assert_zero', 15), ('free', 0, 15)]
    for w0, i1 in ((7, 9), (P + 3, 2), (200, 255), (P, P), (0, 2 ** 32 - 1)):
        wit = [w0, (w0 ^ i1) % P, (i1 & w0) % P, (w0 * w0) % P]
        mine, ref, noncanon = _run(P, gates, [i1], wit, gateset='arithmetic,boolean')
        assert mine == ref == [] and not noncanon, (w0, i1)
        mine, ref, noncanon = _run(P, gates, [i1], [w0, (w0 % P ^ i1 % P) % P + 1, wit[2], wit[3]], gateset='arithmetic,boolean')
        assert mine == ref != [] and not noncanon, (w0, i1)
    # a value wider than the limbs cannot be represented: the lane is flagged
    _, _, noncanon = _run(P, gates, [1], [2 ** 64 + 1, 0, 0, 1], gateset='arithmetic,boolean', width=16)
    assert noncanon
    # GF(2): xor / and only look at the low bit -- a witness byte of 2 or 3 is its residue
    gates = [('witness', 0), ('witness', 1), ('xor', 2, 0, 1), ('and', 3, 2, 1), ('instance', 4), ('xor', 5, 3, 4), ('assert_zero', 5),
             ('free', 0, 5)]
    for w0, w1, e in ((2, 1, 1), (3, 1, 0), (2, 3, 1), (0, 1, 1)):
        mine, ref, noncanon = _run(2, gates, [e], [w0, w1], gateset='boolean', width=1)
        assert not noncanon and mine == ref == [], (w0, w1, e)
    # ... but `not` is `is_zero ? 1 : 0` on the integer: not(2) = 0 for the reference, not(0) = 1.  A position read by
    # zero tests alone is packed as `v != 0`
    gates = [('witness', 0), ('not', 1, 0), ('assert_zero', 1), ('free', 0, 1)]
    for w in (0, 1, 2, 3, 254):
        mine, ref, noncanon = _run(2, gates, [], [w], gateset='boolean', width=1)
        assert not noncanon and mine == ref and (ref == []) == (w != 0), w
        assert _run.modes == ([], [0x01])
    gates = [('witness', 0), ('copy', 1, 0), ('assert_zero', 1), ('free', 0, 1)]
    for w in (0, 1, 2):
        mine, ref, noncanon = _run(2, gates, [], [w], gateset='boolean', width=1)
        assert not noncanon and mine == ref and (ref == []) == (w == 0), w
    # one bit cannot be `v & 1` for a gate and `v != 0` for a zero test: such a position is strict, a value > 1 refused
    gates = [('witness', 0), ('witness', 1), ('not', 2, 0), ('xor', 3, 0, 1), ('xor', 4, 2, 3), ('assert_zero', 4), ('free', 0, 4)]
    for w0, w1 in ((0, 1), (1, 0), (1, 1), (0, 0)):
        mine, ref, noncanon = _run(2, gates, [], [w0, w1], gateset='boolean', width=1)
        assert not noncanon and mine == ref, (w0, w1)
        assert _run.modes == ([], [0xFF, 0x00])
    _, ref, noncanon = _run(2, gates, [], [2, 1], gateset='boolean', width=1)
    assert noncanon


def test_constants_beyond_the_characteristic():
    mod_le = sw.int_to_le(P)
    # in arithmetic: reduced, like the reference's `% m`
    gates = [('witness', 0), ('constant', 1, sw.int_to_le(P + 7)), ('mul', 2, 0, 1), ('addc', 3, 2, sw.int_to_le(P - 21)), ('assert_zero', 3),
             ('free', 0, 3)]
    mine, ref, noncanon = _run(P, gates, [], [3])          # 3 * 7 - 21
    assert mine == ref == [] and not noncanon
    # copied into an assert: the reference sees the integer 101, not 0 -- and so does the assert's entry
    gates = [('constant', 0, sw.int_to_le(P)), ('copy', 1, 0), ('assert_zero', 1), ('free', 0, 1)]
    mine, ref, noncanon = _run(P, gates, [], [])
    assert mine == ref == ['Wire_1 (may be weighted) should be 0, while it is not'] and not noncanon
    gates = [('constant', 0, sw.int_to_le(2 * P)), ('not', 1, 0), ('assert_zero', 1), ('free', 0, 1)]     # not(2p) = 0
    mine, ref, noncanon = _run(P, gates, [], [], gateset='arithmetic,boolean')
    assert mine == ref == [] and not noncanon
    # the same constant in arithmetic AND in front of a zero test
    gates = [('witness', 0), ('constant', 1, sw.int_to_le(P + 7)), ('mul', 2, 0, 1), ('addc', 3, 2, sw.int_to_le(P - 21)), ('assert_zero', 3),
             ('not', 4, 1), ('assert_zero', 4), ('free', 0, 4)]
    mine, ref, noncanon = _run(P, gates, [], [3])
    assert mine == ref == [] and not noncanon
    # GF(2): a constant 2 in front of zero tests alone is `non-zero`; feeding a gate as well it is refused at finalize
    gates = [('constant', 0, bytes([2])), ('copy', 1, 0), ('assert_zero', 1), ('free', 0, 1)]
    mine, ref, noncanon = _run(2, gates, [], [], gateset='boolean', width=1)
    assert mine == ref == ['Wire_1 (may be weighted) should be 0, while it is not']
    gates = [('constant', 0, bytes([2])), ('not', 1, 0), ('assert_zero', 1), ('free', 0, 1)]
    mine, ref, noncanon = _run(2, gates, [], [], gateset='boolean', width=1)
    assert mine == ref == []
    rel = sw.write_relation(bytes([2]), 'boolean', 'simple', [], [('constant', 0, bytes([2])), ('witness', 1), ('not', 2, 0), ('xor', 3, 0, 1),
                                                                 ('free', 0, 3)])
    ev = zk.Evaluator()
    ev.declare_inputs(0, 1)
    ev.ingest_message(rel)
    with pytest.raises(zk.ZkGpuError, match='constant >= the field characteristic reaches assert_zero / not and, as its low bit, a gate'):
        ev.finalize()
    # a constant >= p in an integer bit operation over an odd field: the entry reads the constant as the integer it is
    gates = [('constant', 0, sw.int_to_le(P + 6)), ('witness', 1), ('and', 2, 0, 1), ('copy', 3, 0), ('xor', 4, 3, 1), ('witness', 5), ('witness', 6),
             ('mulc', 7, 5, sw.int_to_le(P - 1)), ('add', 8, 2, 7), ('assert_zero', 8),
             ('mulc', 9, 6, sw.int_to_le(P - 1)), ('add', 10, 4, 9), ('assert_zero', 10), ('free', 0, 10)]
    for w in (5, 100, P + 9, 255):
        mine, ref, noncanon = _run(P, gates, [], [w, ((P + 6) & w) % P, ((P + 6) ^ w) % P], gateset='arithmetic,boolean')
        assert mine == ref == [] and not noncanon, w
        mine, ref, noncanon = _run(P, gates, [], [w, ((P + 6) & w) % P, ((P + 6) % P ^ w % P) % P + 1], gateset='arithmetic,boolean')
        assert mine == ref and (ref != [] or ((P + 6) ^ w) % P == ((P + 6) % P ^ w % P) % P + 1), w
    # ... unless it is wider than the limbs of the field: refused at finalize
    rel = sw.write_relation(mod_le, 'arithmetic,boolean', 'simple', [], [('constant', 0, sw.int_to_le(2 ** 70 + 6)), ('witness', 1), ('and', 2, 0, 1),
                                                                        ('free', 0, 2)])
    ev = zk.Evaluator()
    ev.declare_inputs(0, 1)
    ev.ingest_message(rel)
    with pytest.raises(zk.ZkGpuError, match='constant wider than the field'):
        ev.finalize()
    # a wire left alive at the end that is a constant >= p: Evaluator::get returns the integer (Schedule::raw_source, on the
    # GPU: tests/test_gpu_parity.py)
    rel = sw.write_relation(mod_le, 'arithmetic', 'simple', [], [('constant', 0, sw.int_to_le(P + 1))])
    ev = zk.Evaluator()
    ev.ingest_message(rel)
    ev.finalize()
    assert ev.host_violations() == []


def test_a_wire_alive_at_the_end_is_read_from_its_input():
    # Evaluator::get returns the unreduced integer (evaluator.rs:750-752): a wire nobody freed that is a copy of an input is
    # answered from the input itself (Schedule::raw_source; on the GPU: tests/test_gpu_parity.py), the verdict is unaffected
    gates = [('witness', 0), ('copy', 1, 0), ('mul', 2, 0, 0)]
    for w in (5, P + 5):
        mine, ref, noncanon = _run(P, gates, [], [w])
        assert mine == ref == [] and not noncanon and _run.modes == ([], [0x03])


def test_the_verdict_does_not_depend_on_how_the_tape_was_cut_into_windows():
    """a window cannot know the later readers of a value its owner still holds, so what a value >= p means for a position
    is decided when the tape has ended (Schedule::strict_*), not in the entries a streamed window uploaded long before"""
    chain = [('witness', 0)] + [('addc', k, k - 1, bytes([1])) for k in range(1, 40)] + [('mulc', 40, 0, bytes([2]))]
    cases = {
        'arithmetic only': (chain + [('free', 0, 40)], [P + 1], 0x00),
        'zero test in a later window': (chain + [('copy', 41, 0), ('assert_zero', 41), ('free', 0, 41)], [P], 0x02),
        'alive at the end': (chain + [('free', 1, 40)], [P + 1], 0x03),
    }
    for name, (gates, wit, mode) in cases.items():
        rel = sw.write_relation(sw.int_to_le(P), 'arithmetic', 'simple', [], gates)
        ref = oracle_lane(sw.int_to_le(P), [], wit, [rel], 32, trace=False)
        seen = {}
        for stream in ('0', '16'):
            ev = zk.Evaluator()
            ev.set_option('stream', stream)
            ev.declare_inputs(0, 1)
            ev.ingest_message(rel)
            ev.finalize()
            assert (ev.stream_info()['windows'] > 1) == (stream == '16')
            ops, launches, consts, _ = ev.schedule_dump()
            info = ev.schedule_info()
            modes = (ev.input_modes(False), ev.input_modes(True))
            assert modes == ([], [mode]), (name, stream)
            _, ff, flagged = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], P, [], wit, modes=modes)
            seen[stream] = (expected_product_violations(ev, ff), flagged)
        assert seen['0'] == seen['16'], name
        assert seen['0'] == (ref.violations, False), name


# ---- random relations with inputs at and above the characteristic -----------------------------------------------------
FUZZ_FIELDS = [(P, True), (P, False), (2 ** 61 - 1, True), (2 ** 64 - 2, True), (2 ** 61 - 1, False), (6, True)]


def _unreduced_rows(g, lanes, seed, p):
    """instance / witness rows of 8-byte values: small ones, multiples of p, p + small, anything below 2^64"""
    import random
    r = random.Random(seed)

    def val():
        k = r.random()
        if k < 0.25:
            return r.randrange(0, 6)
        if k < 0.5:
            return min(2 ** 64 - 1, p * r.randrange(1, 4) + r.randrange(0, 3))
        if k < 0.75:
            return r.randrange(p)
        return r.getrandbits(64)
    return [[val() for _ in range(g.n_inst)] for _ in range(lanes)], [[val() for _ in range(g.n_wit)] for _ in range(lanes)]


@pytest.mark.parametrize('seed', range(30))
def test_random_relations_with_unreduced_inputs_against_oracle(seed):
    """structured relations (functions, loops, switches) whose inputs are NOT canonical: arithmetic reduces them, zero tests
    and bit operations see the integers -- the product's verdict is the oracle's for every lane, nobody is flagged"""
    from random_circuits import Gen
    p, boolean = FUZZ_FIELDS[seed % len(FUZZ_FIELDS)]
    g = Gen(seed + 900, p, boolean)
    rel, mod_le = g.relation()
    rows_i, rows_w = _unreduced_rows(g, 4, seed, p)
    ev = zk.Evaluator()
    ev.declare_inputs(g.n_inst, g.n_wit)
    ev.ingest_message(rel)
    if not ev.n_value_ops and ev.host_violations():
        return
    ev.finalize()
    assert ev.elem_bytes == 8
    ops, launches, consts, _ = ev.schedule_dump()
    info = ev.schedule_info()
    modes = (ev.input_modes(False), ev.input_modes(True))
    assert 0xFF not in modes[0] + modes[1]
    for lane in range(4):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32, trace=False)
        _, ff, flagged = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p, rows_i[lane], rows_w[lane],
                                              shuffle_seed=seed, modes=modes)
        assert not flagged and expected_product_violations(ev, ff) == ref.violations, (seed, lane)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(0, 30, 2))
def test_random_relations_with_unreduced_inputs_on_gpu(seed):
    from helpers import batch_arrays
    from random_circuits import Gen
    p, boolean = FUZZ_FIELDS[seed % len(FUZZ_FIELDS)]
    g = Gen(seed + 900, p, boolean)
    rel, mod_le = g.relation(n_top=14)
    lanes = 70
    rows_i, rows_w = _unreduced_rows(g, lanes, seed, p)
    ev = zk.Evaluator()
    ev.declare_inputs(g.n_inst, g.n_wit)
    ev.ingest_message(rel)
    if not ev.n_value_ops and ev.host_violations():
        return
    ev.finalize()
    inst, wit = batch_arrays(rows_i, rows_w, ev.elem_bytes)
    ev.set_inputs(inst if g.n_inst else None, wit if g.n_wit else None, lanes)
    ev.replay()
    ev.synchronize()
    n_ok = 0
    for lane in range(lanes):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32, trace=False)
        assert ev.get_violations(lane) == ref.violations, (seed, lane)
        n_ok += ref.violations == []
    assert ev.counts() == (n_ok, lanes - n_ok) and not ev.lane_results(lanes)[1].any()
    # Evaluator::get on the wires still alive at the end: the integers, reduced or not
    ref = oracle_lane(mod_le, rows_i[0], rows_w[0], [rel], 32, trace=False)
    if not ref.violations:
        for wid in range(0, 60):
            want = ref.get(wid)
            got = ev.get(wid, lanes)
            assert (want is None) == (got is None), (seed, wid)
            if want is not None:
                assert got[0] == want, (seed, wid)
