"""Values >= the field characteristic.  PlaintextBackend keeps constants, instance and witness
values unreduced (rust/src/consumers/evaluator.rs:862-864,896-898,940-946): add / mul / add_constant / mul_constant
reduce their result, but `copy` clones the integer as it is, `assert_zero` and `not` test it for zero, `and` / `xor`
work on its bits and `Evaluator::get` returns it.  The product reduces such a value exactly where the reference's own
arithmetic would (anything that first meets an arithmetic gate; over GF(2) also and / xor, whose low bit only depends
on the low bits), so those statements get the reference's verdict; a value that could reach one of the other consumers
through copies alone is refused -- per lane for inputs, at finalize for constants -- and the refusal says so."""
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
    _, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p, inst, wit)
    ref = oracle_lane(mod_le, inst, wit, [rel], max(width, 32), trace=False)
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


def test_an_unreduced_input_that_reaches_assert_zero_through_copies_is_refused_per_lane():
    # assert_zero(copy(copy(w))): the reference tests the unreduced integer: w = p is NOT zero for it
    gates = [('witness', 0), ('copy', 1, 0), ('copy', 2, 1), ('assert_zero', 2)]
    mine, ref, noncanon = _run(P, gates, [], [0])
    assert mine == ref == [] and not noncanon
    mine, ref, noncanon = _run(P, gates, [], [P])
    assert ref == ['Wire_2 (may be weighted) should be 0, while it is not']      # p != 0 as an integer
    assert noncanon                                                              # the product flags the lane instead of guessing
    # the same input through an arithmetic gate first: p + 0 = 0, both say TRUE
    gates = [('witness', 0), ('addc', 1, 0, bytes([0])), ('assert_zero', 1), ('free', 0, 1)]
    mine, ref, noncanon = _run(P, gates, [], [P])
    assert mine == ref == [] and not noncanon


def test_bit_operations_and_not_are_strict_over_an_odd_field_but_and_xor_are_not_over_gf2():
    # odd p: (w & 3) on the unreduced integer
    gates = [('witness', 0), ('constant', 1, bytes([3])), ('and', 2, 0, 1), ('addc', 3, 2, sw.int_to_le(P - 1)), ('assert_zero', 3)]
    mine, ref, noncanon = _run(P, gates, [], [5], gateset='arithmetic,boolean')      # 5 & 3 = 1
    assert mine == ref == [] and not noncanon
    _, ref, noncanon = _run(P, gates, [], [P + 5], gateset='arithmetic,boolean')      # (106 & 3) = 2 for the reference
    assert noncanon and ref != []
    # GF(2): xor / and only look at the low bit -- a witness byte of 2 or 3 is its residue
    gates = [('witness', 0), ('witness', 1), ('xor', 2, 0, 1), ('and', 3, 2, 1), ('instance', 4), ('xor', 5, 3, 4), ('assert_zero', 5),
             ('free', 0, 5)]
    for w0, w1, e in ((2, 1, 1), (3, 1, 0), (2, 3, 1), (0, 1, 1)):
        mine, ref, noncanon = _run(2, gates, [e], [w0, w1], gateset='boolean', width=1)
        assert not noncanon and mine == ref == [], (w0, w1, e)
    # ... but `not` is `is_zero ? 1 : 0` on the integer: not(2) = 0 for the reference, not(0) = 1
    gates = [('witness', 0), ('not', 1, 0), ('assert_zero', 1)]
    mine, ref, noncanon = _run(2, gates, [], [1], gateset='boolean', width=1)
    assert mine == ref == [] and not noncanon
    mine, ref, noncanon = _run(2, gates, [], [2], gateset='boolean', width=1)
    assert ref == [] and noncanon            # reference: not(2) = 0, TRUE; the residue would give not(0) = 1: refused


def test_constants_beyond_the_characteristic():
    mod_le = sw.int_to_le(P)
    # in arithmetic: reduced, like the reference's `% m`
    gates = [('witness', 0), ('constant', 1, sw.int_to_le(P + 7)), ('mul', 2, 0, 1), ('addc', 3, 2, sw.int_to_le(P - 21)), ('assert_zero', 3),
             ('free', 0, 3)]
    mine, ref, noncanon = _run(P, gates, [], [3])          # 3 * 7 - 21
    assert mine == ref == [] and not noncanon
    # copied into an assert: the reference sees the integer 101, not 0 -- refused at finalize, with the reason
    rel = sw.write_relation(mod_le, 'arithmetic', 'simple', [], [('constant', 0, sw.int_to_le(P)), ('copy', 1, 0), ('assert_zero', 1)])
    ev = zk.Evaluator()
    ev.ingest_message(rel)
    with pytest.raises(zk.ZkGpuError, match='constant >= the field characteristic reaches copy / assert_zero'):
        ev.finalize()
    assert oracle_lane(mod_le, [], [], [rel], 32, trace=False).violations == ['Wire_1 (may be weighted) should be 0, while it is not']
    # a wire left alive at the end can be asked for with Evaluator::get: its constant must be canonical too
    rel = sw.write_relation(mod_le, 'arithmetic', 'simple', [], [('constant', 0, sw.int_to_le(P + 1))])
    ev = zk.Evaluator()
    ev.ingest_message(rel)
    with pytest.raises(zk.ZkGpuError, match='Evaluator::get'):
        ev.finalize()


def test_streamed_windows_are_cautious_about_values_that_stay_open():
    """a window cannot know the later readers of a value its owner still holds: such an input stays strict"""
    gates = [('witness', 0)] + [('addc', k, k - 1, bytes([1])) for k in range(1, 40)] + [('mulc', 40, 0, bytes([2])), ('free', 0, 40)]
    rel = sw.write_relation(sw.int_to_le(P), 'arithmetic', 'simple', [], gates)
    flags = {}
    for stream in ('0', '16'):
        ev = zk.Evaluator()
        ev.set_option('stream', stream)
        ev.declare_inputs(0, 1)
        ev.ingest_message(rel)
        ev.finalize()
        ops, launches, consts, _ = ev.schedule_dump()
        info = ev.schedule_info()
        _, _, flags[stream] = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], P, [], [P + 1])
    assert flags == {'0': False, '16': True}      # one window sees every reader (all arithmetic); the first of three does not
