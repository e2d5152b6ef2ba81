"""A relation whose field characteristic changes between Relation messages.  The reference takes the modulus afresh from
every message header (rust/src/consumers/evaluator.rs:232-237, :262-268): later messages compute under the new modulus and
the wires alive in the scope live on as the integers they are.  The product opens a new FIELD SEGMENT (backend, schedule,
engine per field); the wires of the scope become `carry` inputs of the next segment -- unreduced integers, with the same
rules as any value >= p (tests/test_unreduced.py).  CPU tier: recording + scheduling of every segment, interpreted by
program_sim segment after segment, against the oracle's run of the same messages.  GPU tier: the same through the kernels."""
import numpy as np
import pytest

import circuits
import program_sim
from helpers import batch_arrays, oracle_lane
from test_fuzz_host import expected_product_violations
import zkinterface_ir_amd as zk
from zkinterface_ir_amd import sieve_writer as sw

P1, P2, P3 = 101, 2 ** 61 - 1, 97


def _messages(parts):
    """parts: [(modulus, gateset, gates), ...] -> one Relation message each"""
    return [sw.write_relation(sw.int_to_le(p), gateset, 'simple', [], gates) for p, gateset, gates in parts]


def _session(msgs, n_inst, n_wit, retain=False):
    ev = zk.Evaluator()
    ev.declare_inputs(n_inst, n_wit)
    for m in msgs:
        ev.ingest_message(m)
    ev.finalize(retain_all=retain)
    return ev


def _simulate(ev, moduli, inst, wit):
    """program_sim over the field segments in order; returns (first failing assert or None, flagged)"""
    assert ev.n_field_segments == len(moduli)
    carries, first_fail, flagged = [], None, False
    for k, p in enumerate(moduli):
        ev.set_option('inspect_segment', str(k))
        ops, launches, consts, _ = ev.schedule_dump()
        info = ev.schedule_info()
        seg = ev.field_segment_info(k)
        assert seg['carried_in'] == len(carries) and ev.modulus_le() == sw.int_to_le(p)
        modes = (ev.input_modes(False), ev.input_modes(True), ev.input_modes(2))
        canonical = ev.field_representation(k) == 2   # the any-modulus kernels (tests/test_any_modulus.py)
        slots, ff, nc = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p, inst, wit,
                                             modes=modes, carries=carries, canonical=canonical)
        flagged = flagged or nc
        if ff is not None and (first_fail is None or ff < first_fail):
            first_fail = ff
        if k + 1 < len(moduli):
            carries = [program_sim.from_device_form(slots[sl], p, info['words_per_const'], canonical) for sl in ev.field_segment_carried(k)]
    ev.set_option('inspect_segment', '')
    return first_fail, flagged


# w0 * w1 computed over GF(101), then checked against an instance value over another field, with w0 carried along too
def _two_fields(p_second, epilogue):
    return [(P1, 'arithmetic', [('witness', 0), ('witness', 1), ('mul', 2, 0, 1), ('free', 1, 1)]),
            (p_second, 'arithmetic', epilogue)]


GROW = _two_fields(P2, [('instance', 3), ('mulc', 4, 3, sw.int_to_le(P2 - 1)), ('add', 5, 2, 4), ('assert_zero', 5),
                        ('mul', 6, 0, 0), ('instance', 7), ('mulc', 8, 7, sw.int_to_le(P2 - 1)), ('add', 9, 6, 8), ('assert_zero', 9),
                        ('free', 0, 0), ('free', 2, 9)])
SHRINK = _two_fields(P3, [('instance', 3), ('mulc', 4, 3, sw.int_to_le(P3 - 1)), ('add', 5, 2, 4), ('assert_zero', 5),
                          ('copy', 6, 0), ('assert_zero', 6), ('free', 0, 0), ('free', 2, 6)])


def _lanes_grow():
    # (instances, witnesses): e0 = (w0 * w1 mod 101) as an integer, e1 = w0^2 mod p2
    rows = []
    for w0, w1, ok in ((3, 4, True), (100, 100, True), (7, 50, False), (0, 9, True)):
        e0 = (w0 * w1) % P1 + (0 if ok else 1)
        rows.append(([e0, (w0 * w0) % P2], [w0, w1]))
    return rows


def test_the_modulus_may_grow_between_relation_messages():
    msgs = _messages(GROW)
    ev = _session(msgs, 2, 2)
    assert ev.n_field_segments == 2 and ev.elem_bytes == 8
    # the product travels through the carry stream; w0, still the witness it started as, is read again by the new segment
    assert ev.field_segment_info(0) == {'carried_in': 0, 'assert_base': 0, 'words': 2, 'carried_out': 1}
    assert ev.field_segment_info(1)['carried_in'] == 1
    assert ev.n_asserts == 2 and ev.assert_wires().tolist() == [5, 9]
    for inst, wit in _lanes_grow():
        ref = oracle_lane(sw.int_to_le(P1), inst, wit, msgs, 32, trace=False)
        ff, flagged = _simulate(ev, [P1, P2], inst, wit)
        assert not flagged and expected_product_violations(ev, ff) == ref.violations, (inst, wit)
        assert ref.violations in ([], ['Wire_5 (may be weighted) should be 0, while it is not'])
    assert oracle_lane(sw.int_to_le(P1), *_lanes_grow()[2], msgs, 32, trace=False).violations == ['Wire_5 (may be weighted) should be 0, while it is not']


def test_a_smaller_modulus_sees_the_carried_integers_unreduced():
    """values carried into GF(97) from GF(101) may be >= 97: arithmetic reduces them, assert_zero(copy(w0)) tests the integer"""
    msgs = _messages(SHRINK)
    ev = _session(msgs, 1, 2)
    assert ev.n_field_segments == 2
    ev.set_option('inspect_segment', '1')
    assert ev.input_modes(2) == [0x00]            # the product: arithmetic
    assert ev.input_modes(True) == [0x01, 0x00]   # w0, read again under GF(97): by a zero test alone
    ev.set_option('inspect_segment', '')
    for w0, w1, e0 in ((0, 5, 0), (97, 1, 0), (97, 1, 97), (100, 1, 3), (100, 1, 100), (0, 0, 0), (98, 99, (98 * 99) % 101)):
        inst, wit = [e0], [w0, w1]
        ref = oracle_lane(sw.int_to_le(P1), inst, wit, msgs, 32, trace=False)
        ff, flagged = _simulate(ev, [P1, P3], inst, wit)
        assert not flagged and expected_product_violations(ev, ff) == ref.violations, (inst, wit)
        assert all(v.startswith('Wire_') for v in ref.violations)


def test_three_segments_and_a_wire_alive_at_the_end():
    parts = [(P1, 'arithmetic', [('witness', 0), ('addc', 1, 0, bytes([5]))]),
             (P2, 'arithmetic', [('mul', 2, 1, 1), ('free', 0, 1)]),
             (P1, 'arithmetic', [('instance', 3), ('mulc', 4, 3, sw.int_to_le(P1 - 1)), ('add', 5, 2, 4), ('assert_zero', 5), ('free', 3, 5)])]
    msgs = _messages(parts)
    ev = _session(msgs, 1, 1)
    assert ev.n_field_segments == 3 and [ev.field_segment_info(k)['carried_out'] for k in range(3)] == [1, 1, 0]
    for w, good in ((3, True), (100, True), (50, False)):
        sq = ((w + 5) % P1) ** 2 % P2           # wire 2: an integer < p2, used over GF(101) again
        inst, wit = [(sq + (0 if good else 1)) % P1], [w]
        ref = oracle_lane(sw.int_to_le(P1), inst, wit, msgs, 32, trace=False)
        ff, flagged = _simulate(ev, [P1, P2, P1], inst, wit)
        # wire 2 is alive at the end and may be >= 101: Evaluator::get returns the integer, zkgpu_get_wire reads it from the
        # carry stream (Schedule::raw_source); the verdict does not depend on it
        assert not flagged and expected_product_violations(ev, ff) == ref.violations, w


# Inputs and constants that the old segment has only copied are still the integers they started as when the field changes
# (PlaintextBackend never reduces them, evaluator.rs:862-864,896-898,940-946), possibly >= the old characteristic: the new
# segment reads the input / constant itself (capi.cpp switch_field) instead of a residue from the old wire table.
UNREDUCED = [(P1, 'arithmetic', [('witness', 0), ('copy', 1, 0), ('constant', 2, sw.int_to_le(P1 + 7)), ('copy', 3, 2), ('instance', 4),
                                 ('free', 0, 0), ('free', 2, 2)]),
             (P2, 'arithmetic', [('mul', 5, 1, 1), ('mul', 6, 3, 4), ('add', 7, 5, 6), ('instance', 8), ('mulc', 9, 8, sw.int_to_le(P2 - 1)),
                                 ('add', 10, 7, 9), ('assert_zero', 10), ('copy', 11, 1), ('free', 1, 1), ('free', 3, 10)]),
             (P3, 'arithmetic', [('instance', 12), ('mulc', 13, 12, sw.int_to_le(P3 - 1)), ('add', 14, 11, 13), ('assert_zero', 14),
                                 ('free', 11, 14)])]


def _lanes_unreduced():
    rows = []
    for w0, i4, ok in ((5, 3, True), (150, 250, True), (P1, P1 + 1, True), (2 ** 40 + 1, 2 ** 50, True), (150, 250, False), (P3 + 1, 0, True)):
        e = (w0 * w0 + (P1 + 7) * i4) % P2          # the integers, not their residues mod 101
        rows.append(([i4, e if ok else (e + 1) % P2, w0 % P3], [w0]))
    return rows


def test_inputs_and_constants_cross_a_field_change_as_the_integers_they_are():
    msgs = _messages(UNREDUCED)
    ev = _session(msgs, 3, 1)
    assert ev.n_field_segments == 3 and [ev.field_segment_info(k)['carried_out'] for k in range(3)] == [0, 0, 0]
    for inst, wit in _lanes_unreduced():
        ref = oracle_lane(sw.int_to_le(P1), inst, wit, msgs, 32, trace=False)
        ff, flagged = _simulate(ev, [P1, P2, P3], inst, wit)
        assert not flagged and expected_product_violations(ev, ff) == ref.violations, (inst, wit)
    assert [oracle_lane(sw.int_to_le(P1), i, w, msgs, 32, trace=False).violations == [] for i, w in _lanes_unreduced()] == [True, True, True, True, False, True]


@pytest.mark.gpu
def test_inputs_and_constants_cross_a_field_change_on_the_gpu():
    msgs = _messages(UNREDUCED)
    rows = _lanes_unreduced()
    ev = _session(msgs, 3, 1)
    inst, wit = batch_arrays([r[0] for r in rows], [r[1] for r in rows], ev.elem_bytes)
    ev.set_inputs(inst, wit, len(rows))
    ev.replay()
    ev.synchronize()
    for lane, (iv, wv) in enumerate(rows):
        ref = oracle_lane(sw.int_to_le(P1), iv, wv, msgs, 32, trace=False)
        assert ev.get_violations(lane) == ref.violations, lane
    assert ev.counts() == (5, 1) and not ev.lane_results(len(rows))[1].any()


def test_what_a_field_change_still_refuses():
    # (between GF(2) and another field: tests/test_any_modulus.py -- the GF(2) wires of such a session are integers;
    # through the trait-level entry points: test_field_change_through_the_trait_level_entry_points below)
    # a finalized session records nothing more, whatever the modulus
    ev = zk.Evaluator()
    ev.backend_set_field(bytes([101]))
    ev.backend_assert_zero(ev.backend_witness(0), 0)
    ev.finalize()
    with pytest.raises(zk.ZkGpuError, match='already finalized'):
        ev.backend_set_field(bytes([97]))
    # the gate set alone may change from message to message (it only selects the Evaluator's own arms)
    msgs = [sw.write_relation(bytes([101]), 'arithmetic', 'simple', [], [('witness', 0)]),
            sw.write_relation(bytes([101]), 'arithmetic,boolean', 'simple', [], [('not', 1, 0), ('assert_zero', 1), ('free', 0, 1)])]
    ev = _session(msgs, 0, 1)
    assert ev.n_field_segments == 1 and ev.host_violations() == []


@pytest.mark.gpu
@pytest.mark.parametrize('retain', [False, True])
def test_field_segments_on_the_gpu(retain):
    for parts, moduli, n_inst, rows in (
            (GROW, [P1, P2], 2, _lanes_grow()),
            (SHRINK, [P1, P3], 1, [([e0], [w0, w1]) for w0, w1, e0 in ((0, 5, 0), (97, 1, 0), (97, 1, 97), (100, 1, 3), (100, 1, 100),
                                                                        (98, 99, (98 * 99) % 101))])):
        msgs = _messages(parts)
        ev = _session(msgs, n_inst, 2, retain)
        w = ev.elem_bytes
        inst, wit = batch_arrays([r[0] for r in rows], [r[1] for r in rows], w)
        ev.set_inputs(inst, wit, len(rows))
        ev.replay()
        ev.synchronize()
        n_ok = 0
        for lane, (iv, wv) in enumerate(rows):
            ref = oracle_lane(sw.int_to_le(P1), iv, wv, msgs, 32)
            assert ev.get_violations(lane) == ref.violations, (moduli, lane)
            n_ok += ref.violations == []
            if retain:      # every value-returning backend call of every segment, in call order (the reference stops at its first error)
                # (the wire table holds residues: where the reference's copy keeps an integer >= the new modulus, the dump
                # shows its residue -- what every consumer but a zero test would make of it)
                rv = ref.trace_values()
                exp = [v % moduli[0] for v in rv[:3]] + [v % moduli[1] for v in rv[3:]]     # segment 0 makes three calls
                assert ev.dump_trace_values(len(rows))[lane][:len(rv)] == exp and len(rv) > 4, (moduli, lane)
        assert ev.counts() == (n_ok, len(rows) - n_ok)
        # a second batch through the same chain of engines
        ev.set_inputs(inst, wit, len(rows))
        ev.replay()
        ev.synchronize()
        assert ev.counts() == (n_ok, len(rows) - n_ok)


@pytest.mark.gpu
def test_a_wide_field_after_a_narrow_one_on_the_gpu():
    """GF(101) then BN254: the instance / witness buffers have the width of the widest field (32 bytes per value)"""
    p2 = circuits.BN254_R
    parts = [(P1, 'arithmetic', [('witness', 0), ('witness', 1), ('mul', 2, 0, 1), ('free', 0, 1)]),
             (p2, 'arithmetic', [('witness', 3), ('mul', 4, 2, 3), ('instance', 5), ('mulc', 6, 5, sw.int_to_le(p2 - 1)), ('add', 7, 4, 6),
                                 ('assert_zero', 7), ('free', 2, 7)])]
    msgs = _messages(parts)
    ev = _session(msgs, 1, 3)
    assert ev.elem_bytes == 32
    big = p2 - 12345
    rows = []
    for k in range(70):
        w0, w1 = (7 * k) % P1, (11 * k + 3) % P1
        good = k % 5 != 2
        rows.append(([((w0 * w1) % P1 * big + (0 if good else 1)) % p2], [w0, w1, big]))
    # witnesses of the GF(101) segment far wider than its two words: only arithmetic reads them, so they are reduced, as
    # the reference's `(a * b) % m` does (they used to flag the lane: round-3 advisor finding)
    rows.append(([0], [2 ** 200, 0, 5]))
    rows.append(([((2 ** 200 + 9) * (2 ** 190 + 1) % P1 * 5) % p2], [2 ** 200 + 9, 2 ** 190 + 1, 5]))
    rows.append(([1], [2 ** 255 - 19, 3, 5]))
    inst, wit = batch_arrays([r[0] for r in rows], [r[1] for r in rows], 32)
    ev.set_inputs(inst, wit, len(rows))
    ev.replay()
    ev.synchronize()
    n_ok = 0
    for lane, (iv, wv) in enumerate(rows):
        ref = oracle_lane(sw.int_to_le(P1), iv, wv, msgs, 32, trace=False)
        assert ev.get_violations(lane) == ref.violations, lane
        n_ok += ref.violations == []
    assert ev.counts() == (n_ok, len(rows) - n_ok) and not ev.lane_results(len(rows))[1].any()


def test_functions_and_a_switch_survive_the_field_change():
    """known functions belong to the Evaluator, not to a field: a function declared under GF(101) is called under 2^61 - 1,
    where the Switch weights are exponent ladders over THAT modulus (evaluator.rs:801-839), rewritten by Fermat per segment"""
    functions = [('seg::mul', 1, 2, 0, 0, [('mul', 0, 1, 2)])]
    first = sw.write_relation(sw.int_to_le(P1), 'arithmetic', '@function,@switch,', functions,
                              [('witness', 0), ('witness', 1), ('call', 'seg::mul', [2], [0, 1])])
    second = sw.write_relation(sw.int_to_le(P2), 'arithmetic', '@function,@switch,', [],
                               [('witness', 3),                                          # the switch condition: 0 or 1
                                ('call', 'seg::mul', [4], [2, 2]),                       # (w0 w1 mod 101)^2 over the new field
                                ('switch', 3, [5], [bytes([0]), bytes([1])], [
                                    ('anon', [4, 0], 0, 0, [('add', 0, 1, 2)]),          # case 0: t + w0
                                    ('anon', [4, 1], 0, 0, [('mul', 0, 1, 2)]),          # case 1: t * w1
                                ]),
                                ('instance', 6), ('mulc', 7, 6, sw.int_to_le(P2 - 1)), ('add', 8, 5, 7), ('assert_zero', 8),
                                ('free', 0, 8)])
    msgs = [first, second]
    ev = _session(msgs, 1, 3)
    assert ev.host_violations() == [] and ev.n_field_segments == 2
    ev.set_option('inspect_segment', '1')
    kinds = (ev.schedule_dump()[0][:, 1] & 0xFF).tolist()
    assert kinds.count(13) == 2        # both exponent ladders of the second segment became `x != 0` entries
    ev.set_option('inspect_segment', '')
    for w0, w1, c, good in ((3, 4, 0, True), (3, 4, 1, True), (100, 99, 1, False), (0, 7, 0, True), (5, 5, 2, True)):
        t = ((w0 * w1) % P1) ** 2 % P2
        out = (t + w0) % P2 if c == 0 else (t * w1) % P2 if c == 1 else 0     # no case matches: the weighted sum is 0
        inst, wit = [(out + (0 if good else 1)) % P2], [w0, w1, c]
        ref = oracle_lane(sw.int_to_le(P1), inst, wit, msgs, 32, trace=False)
        ff, flagged = _simulate(ev, [P1, P2], inst, wit)
        assert not flagged and expected_product_violations(ev, ff) == ref.violations, (w0, w1, c)
        assert (ref.violations == []) == good


# ---- fuzz: random structured relations cut into two Relation messages under two moduli ---------------------------------
FIELD_PAIRS = [(P1, P2), (P2, P1), (circuits.BN254_R, P1), (P1, circuits.BN254_R), (P3, P1), (P1, P3), (P2, circuits.P320)]


def _two_field_case(seed):
    """(messages, moduli, generator) or None: a random relation of tests/random_circuits.py whose top-level gates are cut
    at a random place; the functions travel with the first message, constants and inputs are below both moduli, so what is
    >= a modulus is what the first segment hands to a second one over a SMALLER field"""
    from random_circuits import Gen
    import random
    pa, pb = FIELD_PAIRS[seed % len(FIELD_PAIRS)]
    g = Gen(seed, min(pa, pb), False)
    g.relation()
    gates = g.spec['gates']
    cut = random.Random(seed).randrange(1, max(2, len(gates)))
    msgs = [sw.write_relation(sw.int_to_le(pa), 'arithmetic', g.spec['features'], g.spec['functions'], gates[:cut]),
            sw.write_relation(sw.int_to_le(pb), 'arithmetic', g.spec['features'], [], gates[cut:])]
    return msgs, [pa, pb], g


@pytest.mark.parametrize('seed', range(70))
def test_random_relations_under_two_moduli_against_the_oracle(seed):
    msgs, moduli, g = _two_field_case(seed)
    rows_i, rows_w = g.lane_inputs(3, seed + 500)
    ev = zk.Evaluator()
    ev.declare_inputs(g.n_inst, g.n_wit)
    for m in msgs:
        ev.ingest_message(m)
    if ev.n_field_segments == 1:        # the first message already ended in an error: nothing was recorded after it
        assert ev.host_violations() != []
        return
    try:
        ev.finalize()
    except zk.ZkGpuError as e:          # e.g. nothing recorded at all
        assert 'no Relation' in str(e) or 'constant >=' in str(e), e
        return
    for lane in range(3):
        ref = oracle_lane(sw.int_to_le(moduli[0]), rows_i[lane], rows_w[lane], msgs, 32, trace=False)
        ff, flagged = _simulate(ev, moduli, rows_i[lane], rows_w[lane])
        if flagged:                     # a carried integer >= the smaller modulus where its bits / Evaluator::get matter
            continue
        assert expected_product_violations(ev, ff) == ref.violations, (seed, lane)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(0, 70, 3))
def test_random_relations_under_two_moduli_on_the_gpu(seed):
    msgs, moduli, g = _two_field_case(seed)
    lanes = 5
    rows_i, rows_w = g.lane_inputs(lanes, seed + 500)
    ev = zk.Evaluator()
    ev.declare_inputs(g.n_inst, g.n_wit)
    for m in msgs:
        ev.ingest_message(m)
    if ev.n_field_segments == 1:
        return
    try:
        ev.finalize()
    except zk.ZkGpuError:
        return
    inst, wit = batch_arrays(rows_i, rows_w, ev.elem_bytes)
    ev.set_inputs(inst if g.n_inst else None, wit if g.n_wit else None, lanes)
    ev.replay()
    ev.synchronize()
    _, flags = ev.lane_results(lanes)
    for lane in range(lanes):
        if flags[lane]:
            continue
        ref = oracle_lane(sw.int_to_le(moduli[0]), rows_i[lane], rows_w[lane], msgs, 32, trace=False)
        assert ev.get_violations(lane) == ref.violations, (seed, lane)


def test_a_streamed_ingest_gives_up_streaming_at_the_field_change_and_nothing_else():
    """option "stream": the windows scheduled for the first field are dropped at the change (the segment is scheduled at
    finalize like the others), later segments are not streamed; the program is the one a plain ingest makes"""
    chain = [('witness', 0)] + [('addc', k, k - 1, bytes([1])) for k in range(1, 50)] + [('free', 0, 48)]
    msgs = [sw.write_relation(sw.int_to_le(P1), 'arithmetic', 'simple', [], chain),
            sw.write_relation(sw.int_to_le(P2), 'arithmetic', 'simple', [],
                              [('mul', 50, 49, 49), ('instance', 51), ('mulc', 52, 51, sw.int_to_le(P2 - 1)), ('add', 53, 50, 52),
                               ('assert_zero', 53), ('free', 49, 53)])]
    dumps = {}
    for stream in ('0', '16'):
        ev = zk.Evaluator()
        ev.set_option('stream', stream)
        ev.declare_inputs(1, 1)
        for m in msgs:
            ev.ingest_message(m)
        ev.finalize()
        assert ev.n_field_segments == 2 and ev.host_violations() == []
        per_segment = []
        for k in range(2):
            ev.set_option('inspect_segment', str(k))
            ops, launches, consts, _ = ev.schedule_dump()
            per_segment.append((ops.tolist(), launches.tolist(), consts.tolist()))
        ev.set_option('inspect_segment', '')
        dumps[stream] = per_segment
        for w, good in ((3, True), (77, False)):
            e = ((w + 49) % P1) ** 2 % P2 + (0 if good else 5)
            ref = oracle_lane(sw.int_to_le(P1), [e], [w], msgs, 32, trace=False)
            ff, flagged = _simulate(ev, [P1, P2], [e], [w])
            assert not flagged and expected_product_violations(ev, ff) == ref.violations and (ref.violations == []) == good
    assert dumps['0'] == dumps['16']


# ---- the same through the trait-level entry points (zkgpu_backend_*): what a Rust `impl ZKBackend` binds.  The reference's
# Evaluator calls set_field for every Relation message (evaluator.rs:262-268); a new modulus opens a field segment, and the
# wires that live on are the caller's handles that have not been dropped (zkgpu_backend_drop = `impl Drop` of its Wire).
def _trait_session(second_modulus, shrink):
    ev = zk.Evaluator()
    ev.backend_set_field(sw.int_to_le(P1))
    w0 = ev.backend_witness(0)
    w1 = ev.backend_witness(1)
    prod = ev.backend_multiply(w0, w1)
    ev.backend_drop(w1)                                   # ('free', 1, 1)
    ev.backend_set_field(sw.int_to_le(second_modulus))    # the header of the second Relation message
    p2 = second_modulus
    e = ev.backend_instance(0)
    ne = ev.backend_mul_constant(e, sw.int_to_le(p2 - 1))
    d = ev.backend_add(prod, ne)
    c = ev.backend_copy(d)                                # (AssertZero copies its wire, evaluator.rs:340)
    ev.backend_assert_zero(c, 5)
    ev.backend_drop(c)
    if shrink:
        c6 = ev.backend_copy(w0)
        c6b = ev.backend_copy(c6)
        ev.backend_assert_zero(c6b, 6)
        for h in (c6b, c6):
            ev.backend_drop(h)
    else:
        sq = ev.backend_multiply(w0, w0)
        e1 = ev.backend_instance(1)
        ne1 = ev.backend_mul_constant(e1, sw.int_to_le(p2 - 1))
        d1 = ev.backend_add(sq, ne1)
        c1 = ev.backend_copy(d1)
        ev.backend_assert_zero(c1, 9)
        for h in (c1, d1, ne1, e1, sq):
            ev.backend_drop(h)
    for h in (d, ne, e, prod, w0):
        ev.backend_drop(h)
    return ev


@pytest.mark.parametrize('shrink', [False, True])
def test_field_change_through_the_trait_level_entry_points(shrink):
    parts, p2 = (SHRINK, P3) if shrink else (GROW, P2)
    msgs = _messages(parts)
    ev = _trait_session(p2, shrink)
    ev.finalize()
    assert ev.n_field_segments == 2
    assert ev.field_segment_info(0)['carried_out'] == 1 and ev.field_segment_info(1)['carried_in'] == 1
    assert ev.assert_wires().tolist() == ([5, 6] if shrink else [5, 9])
    lanes = [([e0], [w0, w1]) for w0, w1, e0 in ((0, 5, 0), (97, 1, 0), (97, 1, 97), (100, 1, 3), (100, 1, 100), (98, 99, (98 * 99) % 101))] \
        if shrink else _lanes_grow()
    for inst, wit in lanes:
        ref = oracle_lane(sw.int_to_le(P1), inst, wit, msgs, 32, trace=False)
        ff, flagged = _simulate(ev, [P1, p2], inst, wit)
        want = [] if ff is None else ['Wire_%d (may be weighted) should be 0, while it is not' % ev.assert_wires()[ff]]
        assert not flagged and want == ref.violations, (inst, wit, ff)


def test_a_caller_driven_field_change_needs_the_drops_and_keeps_old_handles_valid():
    ev = zk.Evaluator()
    ev.backend_set_field(sw.int_to_le(P1))
    a = ev.backend_witness(0)
    b = ev.backend_add_constant(a, bytes([7]))
    ev.backend_drop(a)
    ev.backend_set_field(sw.int_to_le(P2))
    # a was dropped before the change: it has no value in the new segment; b lives on under its old handle
    with pytest.raises(zk.ZkGpuError, match='dropped before the field characteristic changed'):
        ev.backend_copy(a)
    c = ev.backend_multiply(b, b)
    assert c > b                                           # handles are session-wide: never reused across the change
    ev.backend_assert_zero(ev.backend_copy(c), 1)
    ev.finalize()
    assert ev.n_field_segments == 2 and ev.tape_len == 2 + 3   # witness, add_constant | multiply, copy, assert_zero
    ff, _ = _simulate(ev, [P1, P2], [], [P1 - 7])          # (94 + 7) mod 101 = 0, squared: 0
    assert ff is None
    ff, _ = _simulate(ev, [P1, P2], [], [1])
    assert ff == 0


# ---- a wide field, then a narrow one: the carried values and the inputs are wider than the second field's limbs ----
BN = circuits.BN254_R
WIDE_THEN_NARROW = [(BN, 'arithmetic', [('witness', 0), ('witness', 1), ('mul', 2, 0, 1), ('free', 1, 1)]),
                    (P2, 'arithmetic', [('mul', 3, 2, 0), ('instance', 4), ('mulc', 5, 4, sw.int_to_le(P2 - 1)), ('add', 6, 3, 5),
                                        ('assert_zero', 6), ('free', 0, 0), ('free', 2, 6)])]


def _lanes_wide_then_narrow():
    rows = []
    for w0, w1, ok in ((3, 4, True), (BN - 1, BN - 2, True), (2 ** 200 + 12345, 2 ** 253 + 7, False), (2 ** 64, 2 ** 61 - 1, True), (0, 5, True)):
        prod = (w0 * w1) % BN                      # an integer below BN254's r: far wider than 2^61
        e = (prod * w0) % P2 + (0 if ok else 1)    # the narrow field multiplies the carried product by the witness read again
        rows.append(([e % P2 if ok else e], [w0, w1]))
    return rows


def test_a_narrower_field_reduces_the_wide_values_it_takes_over():
    """BN254 then 2^61 - 1: the carried product (8 words) and the witness read again (8 words in the session's buffers) are
    wider than the second field's two words.  The reference's gates are `(a * b) % m` of whatever integers come in
    (evaluator.rs:908-922): the values are reduced, no lane is flagged (round-3 advisor finding)."""
    msgs = _messages(WIDE_THEN_NARROW)
    ev = _session(msgs, 1, 2)
    assert ev.n_field_segments == 2 and ev.elem_bytes == 32
    ev.set_option('inspect_segment', '1')
    assert ev.input_modes(2) == [0x00] and ev.input_modes(True)[0] == 0x00
    ev.set_option('inspect_segment', '')
    for inst, wit in _lanes_wide_then_narrow():
        ref = oracle_lane(sw.int_to_le(BN), inst, wit, msgs, 32, trace=False)
        ff, flagged = _simulate(ev, [BN, P2], inst, wit)
        assert not flagged and expected_product_violations(ev, ff) == ref.violations, (inst, wit)
    assert [oracle_lane(sw.int_to_le(BN), i, w, msgs, 32, trace=False).violations == [] for i, w in _lanes_wide_then_narrow()] == [True, True, False, True, True]


@pytest.mark.gpu
def test_wide_then_narrow_fields_on_the_gpu():
    msgs = _messages(WIDE_THEN_NARROW)
    rows = _lanes_wide_then_narrow()
    for retain in (False, True):
        ev = _session(msgs, 1, 2, retain)
        inst, wit = batch_arrays([r[0] for r in rows], [r[1] for r in rows], ev.elem_bytes)
        ev.set_inputs(inst, wit, len(rows))
        ev.replay()
        ev.synchronize()
        for lane, (i, w) in enumerate(rows):
            ref = oracle_lane(sw.int_to_le(BN), i, w, msgs, 32, trace=False)
            assert ev.get_violations(lane) == ref.violations, lane
        assert ev.counts() == (4, 1) and not ev.lane_results(len(rows))[1].any()
    # the any-modulus kernels take the same values: an even second modulus
    p_even = 2 ** 61 - 2
    parts = [WIDE_THEN_NARROW[0], (p_even, 'arithmetic', [('mul', 3, 2, 0), ('instance', 4), ('mulc', 5, 4, sw.int_to_le(p_even - 1)), ('add', 6, 3, 5),
                                                          ('assert_zero', 6), ('free', 0, 0), ('free', 2, 6)])]
    msgs2 = _messages(parts)
    rows2 = [([((w0 * w1) % BN * w0) % p_even], [w0, w1]) for _, (w0, w1) in rows]
    ev = _session(msgs2, 1, 2)
    assert ev.field_representation(1) == 2
    inst, wit = batch_arrays([r[0] for r in rows2], [r[1] for r in rows2], ev.elem_bytes)
    ev.set_inputs(inst, wit, len(rows2))
    ev.replay()
    ev.synchronize()
    for lane, (i, w) in enumerate(rows2):
        ref = oracle_lane(sw.int_to_le(BN), i, w, msgs2, 32, trace=False)
        assert ev.get_violations(lane) == ref.violations == [], lane
    assert ev.counts() == (len(rows2), 0)


@pytest.mark.gpu
def test_a_message_value_wider_than_the_buffers_is_reduced_under_the_field_that_reads_it():
    """`evaluate <workspace>` use: GF(101) then GF(2^61 - 1), the witness 2^70 + 3 of the Witness message is wider than the
    session's 8-byte values.  It is multiplied in the FIRST segment: its residue mod 101 goes in, not mod 2^61 - 1 (round-3
    advisor finding); the verdict is the oracle's."""
    big = 2 ** 70 + 3
    for ok in (True, False):
        e = ((big * 7) % P1) ** 2 % P2 + (0 if ok else 1)
        stmt = [sw.write_instance(sw.int_to_le(P1), [sw.int_to_le(e)]),
                sw.write_witness(sw.int_to_le(P1), [sw.int_to_le(big), sw.int_to_le(7)])] + \
            _messages([(P1, 'arithmetic', [('witness', 0), ('witness', 1), ('mul', 2, 0, 1), ('free', 0, 1)]),
                       (P2, 'arithmetic', [('mul', 3, 2, 2), ('instance', 4), ('mulc', 5, 4, sw.int_to_le(P2 - 1)), ('add', 6, 3, 5),
                                           ('assert_zero', 6), ('free', 2, 6)])])
        ev = zk.Evaluator.from_messages(stmt)
        ev.finalize()
        assert ev.n_field_segments == 2 and ev.elem_bytes == 8
        ev.set_inputs_from_messages()
        ev.replay()
        ev.synchronize()
        from oracle_lib import OracleRun
        ref = OracleRun(buffers=stmt, trace=False)
        assert ev.get_violations(0) == ref.violations and (ref.violations == []) == ok
