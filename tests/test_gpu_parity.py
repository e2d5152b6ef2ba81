"""GPU tier: the HIP path, called through the C ABI, against the oracle.

Bar: bit-exact wire values (integer work) and identical violation strings.  Sizes
the oracle finishes in seconds are compared wire by wire; the BASELINE.json size
is checked through size-independent properties (known satisfied count, sampled
lanes against the oracle, replay idempotence, lane-group invariance)."""
import numpy as np
import pytest

import circuits
from helpers import batch_arrays, golden_buffers, oracle_lane
from oracle_lib import OracleRun
import zkinterface_ir_amd as zk
from zkinterface_ir_amd import sieve_writer as sw
from zkinterface_ir_amd import workloads

pytestmark = pytest.mark.gpu

SINGLE = ['ref_examples', 'arith_101_correct', 'arith_101_incorrect', 'bool_correct', 'bool_incorrect',
          'arith_bn254_correct']


@pytest.mark.parametrize('retain', [True, False])
@pytest.mark.parametrize('name', SINGLE)
def test_single_statement_matches_oracle(name, retain):
    """`zki_sieve evaluate <workspace>` for the reference's own statements: verdict, violation
    string and (retain) every backend-op value up to the oracle's stopping point."""
    bufs = golden_buffers(name)
    ev = zk.Evaluator.from_messages(bufs)
    ev.finalize(retain_all=retain)
    ev.set_inputs_from_messages()
    ev.replay()
    ev.synchronize()
    ref = OracleRun(buffers=bufs)
    assert ev.get_violations(0) == ref.violations
    assert ev.get_violations(0) == circuits.GOLDEN_TRACES[name][2]
    sat, failed = ev.counts()
    assert (sat, failed) == ((1, 0) if not ref.violations else (0, 1))
    if retain:
        rv = ref.trace_values()
        assert ev.dump_trace_values(1)[0][:len(rv)] == rv


def test_evaluate_helper_prints_reference_verdicts():
    assert zk.evaluate(golden_buffers('arith_101_correct')) == []
    assert zk.evaluate(golden_buffers('arith_101_incorrect')) == [
        'Wire_9 (may be weighted) should be 0, while it is not']
    inst, wit, rel = golden_buffers('arith_101_correct')
    assert zk.evaluate([inst, wit]) == ['Did not receive any gate to verify.']


def test_valid_eval_metrics_three_part_report(tmp_path):
    """`zki_sieve valid-eval-metrics` (cli.rs:333-363; run on both examples by cli.rs:574-627): validator
    verdict, evaluator verdict (GPU replay) and the Stats JSON, from one pass over the messages."""
    import io
    import json
    from zkinterface_ir_amd import cli
    from validator_ref import StatsRef
    for name, specs, verdict in (('arith', circuits.arith_example_specs(), None),
                                 ('bool', circuits.bool_example_specs(), None),
                                 ('arith_bad', circuits.arith_example_specs(incorrect=True),
                                  'Wire_9 (may be weighted) should be 0, while it is not'),
                                 ('bool_bad', circuits.bool_example_specs(incorrect=True),
                                  'Wire_22 (may be weighted) should be 0, while it is not')):
        d = tmp_path / name
        d.mkdir()
        for k, s in enumerate(specs):
            (d / ('%03d_%s.sieve' % (k, s['type']))).write_bytes(circuits.emit_spec(s))
        want_stats = StatsRef()
        for s in specs:
            want_stats.ingest(s)
        valid, evald, stats = zk.valid_eval_metrics([str(d)])
        assert valid == []
        assert evald == ([verdict] if verdict else [])
        assert json.loads(stats) == want_stats.as_dict()
        err, out = io.StringIO(), io.StringIO()
        rc = cli.main(['valid-eval-metrics', str(d)], err=err, out=out)
        assert json.loads(out.getvalue()) == want_stats.as_dict()
        if verdict:
            assert rc == 1
            assert err.getvalue() == ('\nThe statement is COMPLIANT with the specification!\n'
                                      '\nThe statement is NOT TRUE!\nViolations:\n- %s\n\nError: Found 1 violations.\n' % verdict)
        else:
            assert rc == 0
            assert err.getvalue() == ('\nThe statement is COMPLIANT with the specification!\n'
                                      '\nThe statement is TRUE!\n')


def _batched_example(modulus, lanes):
    """Lane inputs for the arithmetic example relation (examples.rs:72-212): the switch condition
    (witness 0) stays 3 so the pythagorean branch is live; it checks instance0 == 3^2 + witness1^2
    and witness3 == fib(22).  Every third lane gets a wrong instance0."""
    rows_i, rows_w = [], []
    rng = np.random.default_rng(11)
    for lane in range(lanes):
        b = int(rng.integers(0, 2 ** 62)) % modulus
        good = lane % 3 != 1
        rows_i.append([(9 + b * b + (0 if good else 1)) % modulus, 0, 1])
        rows_w.append([3, b, 0, 17711 % modulus])
    return rows_i, rows_w


@pytest.mark.parametrize('modulus', [101, circuits.BN254_R, 2 ** 61 - 1, 2 ** 127 - 1, circuits.P320,
                                     circuits.BLS12_381_Q, circuits.P448, circuits.P512])
def test_batched_example_matches_oracle_per_lane(modulus):
    """One relation, many (instance, witness) pairs: lane i == i-th reference run."""
    lanes = 70  # spans two 64-lane blocks, ragged tail
    _, _, rel = circuits.arith_example(modulus)
    rows_i, rows_w = _batched_example(modulus, lanes)
    ev = zk.Evaluator()
    ev.declare_inputs(3, 4)
    ev.ingest_message(rel)
    assert ev.host_violations() == []
    ev.finalize(retain_all=True)
    w = ev.elem_bytes
    inst, wit = batch_arrays(rows_i, rows_w, w)
    ev.set_inputs(inst, wit, lanes)
    ev.replay()
    ev.synchronize()
    vals = ev.dump_trace_values(lanes)
    mod_le = sw.int_to_le(modulus) if modulus >= 2 ** 32 else modulus.to_bytes(4, 'little')
    n_ok = 0
    for lane in range(lanes):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], w)
        assert ev.get_violations(lane) == ref.violations, lane
        rv = ref.trace_values()
        assert vals[lane][:len(rv)] == rv, lane
        n_ok += not ref.violations
    assert ev.counts() == (n_ok, lanes - n_ok)
    assert 0 < n_ok < lanes
    assert w == 8 * ((modulus.bit_length() + 63) // 64)


@pytest.mark.parametrize('modulus', [101, circuits.BN254_R, 2 ** 61 - 1, circuits.P320, circuits.P512])
def test_switch_weights_in_the_production_schedule(modulus):
    """The compact schedule (exponent ladder of every Switch weight replaced by one `x != 0` entry, copies
    propagated, gates fused) against one oracle run per lane; the lanes take the first branch, the second branch
    or none (condition 3, 5, anything else)."""
    lanes = 96
    _, _, rel = circuits.arith_example(modulus)
    rows_i, rows_w = _batched_example(modulus, lanes)
    for lane in range(lanes):
        if lane % 4 == 1:
            rows_w[lane][0] = 5
        elif lane % 4 == 2:
            rows_w[lane][0] = (7 + lane) % modulus
    ev = zk.Evaluator()
    ev.declare_inputs(3, 4)
    ev.ingest_message(rel)
    ev.finalize()
    from helpers import working_entries
    assert working_entries(ev) < 130
    w = ev.elem_bytes
    inst, wit = batch_arrays(rows_i, rows_w, w)
    ev.set_inputs(inst, wit, lanes)
    ev.replay()
    ev.synchronize()
    mod_le = sw.int_to_le(modulus) if modulus >= 2 ** 32 else modulus.to_bytes(4, 'little')
    n_ok = 0
    for lane in range(lanes):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], w, trace=False)
        assert ev.get_violations(lane) == ref.violations, lane
        n_ok += not ref.violations
    assert ev.counts() == (n_ok, lanes - n_ok)
    off = zk.Evaluator()
    off.set_option('fermat', '0')
    off.declare_inputs(3, 4)
    off.ingest_message(rel)
    off.finalize()
    off.set_inputs(inst, wit, lanes)
    off.replay()
    off.synchronize()
    assert np.array_equal(off.lane_results(lanes)[0], ev.lane_results(lanes)[0])


def test_unreduced_inputs_get_the_reference_verdict():
    """The reference keeps inputs unreduced (evaluator.rs:862-864,940-946).  Where a value >= p first meets an
    arithmetic gate its residue is what the reference computes with too: the kernels reduce it on load and the lane
    gets the oracle's verdict.  Where it reaches assert_zero / not through copies alone it is "not zero", as it is for the
    reference's integer test: the oracle's verdict again.  Where its BITS matter (and / xor over an odd field, Evaluator::get)
    the entry reads the raw input itself (tests/test_unreduced.py has the cases one by one)."""
    _, _, rel = circuits.arith_example(101)
    ev = zk.Evaluator()
    ev.declare_inputs(3, 4)
    ev.ingest_message(rel)
    ev.finalize()
    w = ev.elem_bytes
    rows_i = [[25, 0, 1], [25 + 101, 0, 1], [25, 101, 1], [25, 0, 102], [26 + 101, 0, 1]]
    rows_w = [[3, 4, 0, 36], [3 + 101, 4, 0, 36], [3, 4, 101, 36 + 101], [3, 4, 0, 36], [3, 4, 0, 36]]
    inst, wit = batch_arrays(rows_i, rows_w, w)
    ev.set_inputs(inst, wit, 5)
    ev.replay()
    ev.synchronize()
    n_ok = 0
    for lane in range(5):
        ref = oracle_lane(bytes([101]), rows_i[lane], rows_w[lane], [rel], 4, trace=False)
        assert ev.get_violations(lane) == ref.violations, lane      # every use in this relation is arithmetic
        n_ok += ref.violations == []
    assert n_ok == 4 and ev.counts() == (4, 1)
    # assert_zero(copy(w)): w = p is not zero for the reference's integer test, and the assert's entry tests the raw input
    # beside the wire -- the reference's verdict, lane by lane
    from zkinterface_ir_amd import sieve_writer as sw
    for retain in (False, True):
        rel2 = sw.write_relation(bytes([101]), 'arithmetic,boolean', 'simple', [],
                                 [('witness', 0), ('copy', 1, 0), ('assert_zero', 1), ('witness', 2), ('not', 3, 2), ('mulc', 4, 3, bytes([5])),
                                  ('addc', 5, 4, bytes([96])), ('assert_zero', 5), ('free', 0, 5)])
        ev = zk.Evaluator()
        ev.declare_inputs(0, 2)
        ev.ingest_message(rel2)
        ev.finalize(retain_all=retain)
        assert ev.input_modes(True) == [0x01, 0x01]
        rows_w = [[0, 0], [101, 0], [5, 0], [0, 101], [0, 7], [202, 0], [0, 2 ** 32 - 1], [101, 101]]
        _, wit = batch_arrays([[]] * len(rows_w), rows_w, ev.elem_bytes)
        ev.set_inputs(None, wit, len(rows_w))
        ev.replay()
        ev.synchronize()
        n_true = 0
        for lane, row in enumerate(rows_w):
            ref = oracle_lane(bytes([101]), [], row, [rel2], 4, trace=False)
            assert ev.get_violations(lane) == ref.violations, (retain, lane)
            n_true += ref.violations == []
        assert ev.get_violations(1) == ['Wire_1 (may be weighted) should be 0, while it is not']     # w = p
        assert ev.get_violations(3) == ['Wire_5 (may be weighted) should be 0, while it is not']     # not(p) = 0
        assert ev.counts() == (n_true, len(rows_w) - n_true) and n_true == 1
        assert not ev.lane_results(len(rows_w))[1].any()                                              # nobody flagged
    # and / xor over an odd field work on the bits of the unreduced integer (evaluator.rs:924-933): the entry reads the raw
    # input instead of the wire -- unfused entries (retain_all), fused entries, and the any-modulus kernels (an even p)
    for p in (101, 2 ** 64 - 2):
        neg = sw.int_to_le(p - 1)
        gates = [('witness', 0), ('instance', 1), ('copy', 2, 0), ('copy', 3, 2), ('xor', 4, 3, 1), ('and', 5, 1, 2), ('mul', 6, 0, 0),
                 ('witness', 7), ('witness', 8), ('witness', 9),
                 ('mulc', 10, 7, neg), ('add', 11, 4, 10), ('assert_zero', 11),
                 ('mulc', 12, 8, neg), ('add', 13, 5, 12), ('assert_zero', 13),
                 ('mulc', 14, 9, neg), ('add', 15, 6, 14), ('assert_zero', 15), ('free', 0, 15)]
        rel3 = sw.write_relation(sw.int_to_le(p), 'arithmetic,boolean', 'simple', [], gates)
        top = 2 ** 64 - 1
        rows = [(7, 9), (p + 3 if p + 3 <= top else top, 2), (200, 255), (p, p), (0, 2 ** 32 - 1), (top, p), (top, top - 1)]
        rows_i = [[i1] for _, i1 in rows]
        rows_w = [[w0, (w0 ^ i1) % p, (i1 & w0) % p, (w0 * w0) % p] for w0, i1 in rows]
        rows_i.append([9])
        rows_w.append([7, ((7 ^ 9) + 1) % p, 1, 49 % p])      # a wrong xor: Wire_11
        for retain in (False, True):
            ev = zk.Evaluator()
            ev.declare_inputs(1, 4)
            ev.ingest_message(rel3)
            ev.finalize(retain_all=retain)
            assert ev.input_modes(True)[0] == 0x03 and ev.input_modes(False) == [0x03]
            inst, wit = batch_arrays(rows_i, rows_w, ev.elem_bytes)
            ev.set_inputs(inst, wit, len(rows_w))
            ev.replay()
            ev.synchronize()
            for lane in range(len(rows_w)):
                ref = oracle_lane(sw.int_to_le(p), rows_i[lane], rows_w[lane], [rel3], 8, trace=False)
                assert ev.get_violations(lane) == ref.violations, (p, retain, lane)
                assert (ref.violations == []) == (lane != len(rows_w) - 1)
            assert ev.counts() == (len(rows_w) - 1, 1) and not ev.lane_results(len(rows_w))[1].any()
    # Evaluator::get of a wire that is a copy of an input returns the integer the witness holds (evaluator.rs:750-752)
    rel5 = sw.write_relation(bytes([101]), 'arithmetic', 'simple', [], [('witness', 0), ('copy', 1, 0), ('mul', 2, 0, 0)])
    ev = zk.Evaluator()
    ev.declare_inputs(0, 1)
    ev.ingest_message(rel5)
    ev.finalize()
    _, wit = batch_arrays([[]] * 3, [[5], [106], [2 ** 32 - 1]], ev.elem_bytes)
    ev.set_inputs(None, wit, 3)
    ev.replay()
    ev.synchronize()
    assert ev.get(0, 3) == [5, 106, 2 ** 32 - 1] and ev.get(1, 3) == [5, 106, 2 ** 32 - 1]
    assert ev.get(2, 3) == [25, 25, ((2 ** 32 - 1) ** 2) % 101]
    assert ev.counts() == (3, 0)
    # a constant >= p: its bits in and / xor (the entry reads it from the pool as the integer it is), Evaluator::get of a copy
    for p in (101, 2 ** 64 - 2):
        c = p + 6 if p == 101 else 2 ** 64 - 1
        neg = sw.int_to_le(p - 1)
        gates = [('constant', 0, sw.int_to_le(c)), ('witness', 1), ('and', 2, 0, 1), ('copy', 3, 0), ('xor', 4, 3, 1), ('witness', 5), ('witness', 6),
                 ('mulc', 7, 5, neg), ('add', 8, 2, 7), ('assert_zero', 8), ('mulc', 9, 6, neg), ('add', 10, 4, 9), ('assert_zero', 10),
                 ('free', 0, 2), ('free', 4, 10)]
        rel6 = sw.write_relation(sw.int_to_le(p), 'arithmetic,boolean', 'simple', [], gates)
        ws = [5, 100, p + 9 if p == 101 else 2 ** 63 + 12345, 255, 0]
        rows_w = [[w, (c & w) % p, (c ^ w) % p] for w in ws] + [[7, (c & 7) % p, ((c ^ 7) + 1) % p]]
        for retain in (False, True):
            ev = zk.Evaluator()
            ev.declare_inputs(0, 3)
            ev.ingest_message(rel6)
            ev.finalize(retain_all=retain)
            _, wit = batch_arrays([[]] * len(rows_w), rows_w, ev.elem_bytes)
            ev.set_inputs(None, wit, len(rows_w))
            ev.replay()
            ev.synchronize()
            for lane, row in enumerate(rows_w):
                ref = oracle_lane(sw.int_to_le(p), [], row, [rel6], 8, trace=False)
                assert ev.get_violations(lane) == ref.violations, (p, retain, lane)
            assert ev.counts() == (len(rows_w) - 1, 1)
            assert ev.get(3, len(rows_w)) == [c] * len(rows_w)      # wire 3, a copy of the constant, is alive at the end
    # GF(2): a position only zero tests read is packed as `v != 0` (both kernels)
    rel4 = sw.write_relation(bytes([2]), 'boolean', 'simple', [],
                             [('witness', 0), ('not', 1, 0), ('assert_zero', 1), ('witness', 2), ('copy', 3, 2), ('assert_zero', 3)]
                             + [('witness', 4 + k) for k in range(20)] + [('free', 0, 23)])
    rows_w = [[1, 0] + [0] * 20, [2, 0] + [1] * 20, [0, 0] + [0] * 20, [254, 2] + [1] * 20, [3, 0] + [3] * 20]
    for path in ('hbm', 'lds'):
        ev = zk.Evaluator()
        ev.set_option('bool_path', path)
        ev.declare_inputs(0, 22)
        ev.ingest_message(rel4)
        ev.finalize()
        assert ev.input_modes(True)[:2] == [0x01, 0x01]
        _, wit = batch_arrays([[]] * len(rows_w), rows_w, 1)
        ev.set_inputs(None, wit, len(rows_w))
        ev.replay()
        ev.synchronize()
        for lane, row in enumerate(rows_w):
            ref = oracle_lane(bytes([2]), [], row, [rel4], 1, trace=False)
            assert ev.get_violations(lane) == ref.violations, (path, lane)
        assert ev.counts() == (3, 2)
    # GF(2), Evaluator::get: a wire alive at the end that is a copy of an input returns the byte the witness holds, and a
    # constant 2 alive at the end the integer 2 (the table itself holds one bit per witness)
    rel7 = sw.write_relation(bytes([2]), 'boolean', 'simple', [],
                             [('witness', 0), ('copy', 1, 0), ('witness', 2), ('xor', 3, 0, 2), ('constant', 4, bytes([2])), ('copy', 5, 4)])
    for path in ('hbm', 'lds'):
        ev = zk.Evaluator()
        ev.set_option('bool_path', path)
        ev.declare_inputs(0, 2)
        ev.ingest_message(rel7)
        ev.finalize()
        _, wit = batch_arrays([[]] * 4, [[1, 0], [3, 1], [0, 0], [254, 1]], 1)
        ev.set_inputs(None, wit, 4)
        ev.replay()
        ev.synchronize()
        assert ev.get(1, 4) == [1, 3, 0, 254] and ev.get(0, 4) == [1, 3, 0, 254], path
        assert ev.get(3, 4) == [1, 0, 0, 1], path          # (w0 ^ w2) % 2 on the integers: the low bits
        assert ev.get(5, 4) == [2, 2, 2, 2], path
        assert ev.counts() == (4, 0)


def _layered_session(wl, batch, lane_group=0):
    """The full relation with its epilogue over a batch whose expected outputs come from the CPU (`cpu_opt` on the tape of
    the relation without its epilogue, tests/cpu_checkers.py) -- never from the GPU.  At the BASELINE sizes every lane of
    them is also checked against the committed oracle-chain hashes (tests/golden/c2_all_lanes.json).
    Returns (evaluator, inst, wit, n_bad)."""
    import json
    import os
    import cpu_checkers
    from helpers import ROOT
    inst, wit = wl.inputs(batch)
    outs = cpu_checkers.arith_layered_outputs(wl, inst, wit)
    if (wl.W, wl.D, wl.n_out, wl.seed) == (4096, 256, 64, 0x5EED0001):
        cpu_checkers.check_against_all_lanes_fixture(json.load(open(os.path.join(ROOT, 'tests', 'golden', 'c2_all_lanes.json'))), outs)
    n_bad = wl.set_expected_outputs(inst, outs)
    ev = zk.Evaluator()
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in wl.relation_messages():
        ev.ingest_message(m)
    assert ev.host_violations() == []
    ev.finalize()
    if lane_group:
        ev.set_lane_group(lane_group)
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    return ev, inst, wit, n_bad


def test_layered_relation_small_all_lanes_against_oracle():
    wl = workloads.ArithLayered(W=256, D=12, n_instance0=16, n_out=8)
    batch = 130
    ev, inst, wit, n_bad = _layered_session(wl, batch)
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (batch - n_bad, n_bad)
    msgs = wl.relation_messages()
    for lane in list(range(0, batch, 13)) + [97, batch - 1]:
        iv = [int.from_bytes(inst[lane, k].tobytes(), 'little') for k in range(wl.n_instance)]
        wv = [int.from_bytes(wit[lane, k].tobytes(), 'little') for k in range(wl.n_witness)]
        ref = oracle_lane(wl.mod_le, iv, wv, msgs, wl.width, trace=False)
        assert ev.get_violations(lane) == ref.violations, lane
        assert (ref.violations == []) == (lane % 97 != 0)


def test_full_size_c2_properties():
    """BASELINE.json configs[1]: BN254, 2^20 Add/Mul gates, batch 1024."""
    wl = workloads.ArithLayered()  # W=4096, D=256
    batch = 1024
    ev, inst, wit, n_bad = _layered_session(wl, batch)
    assert ev.n_value_ops == 4096 * 257 + 64 * 3 + 64 and ev.n_asserts == 64
    ev.replay()
    ev.synchronize()
    first = ev.lane_results(batch)
    assert ev.counts() == (workloads.expected_satisfied(batch), n_bad)
    # corrupted lanes fail at their first output comparison: assert #0 -> local wire id of that Add
    bad_lanes = [i for i in range(batch) if i % 97 == 0]
    assert all(first[0][i] == 0 for i in bad_lanes)
    assert all(first[0][i] == zk.NO_FAIL for i in range(batch) if i % 97)
    assert ev.get_violations(97) == ['Wire_%d (may be weighted) should be 0, while it is not' % ((wl.D + 1) * wl.W + 2)]
    # idempotence: a second replay of the same inputs gives the same per-lane words
    ev.replay()
    ev.synchronize()
    again = ev.lane_results(batch)
    assert np.array_equal(first[0], again[0]) and np.array_equal(first[1], again[1])
    # lane groups (Infinity-Cache sized passes) do not change results
    ev.set_lane_group(256)
    ev.replay()
    ev.synchronize()
    grouped = ev.lane_results(batch)
    assert np.array_equal(first[0], grouped[0])
    # every lane against the optimised CPU evaluator (oracle/cpu_opt.cpp; itself pinned to the literal oracle)
    from oracle_lib import opt_eval
    kinds, ta, tb = ev.tape()
    ff_cpu, _, _ = opt_eval(kinds, ta, tb, ev.constants(), wl.mod_le, inst.tobytes(), wl.n_instance, wit.tobytes(),
                            wl.n_witness, wl.width, batch, 16)
    assert np.array_equal(ff_cpu, first[0])
    # two sampled lanes of the full relation against the oracle (a good and a corrupted one)
    msgs = wl.relation_messages()
    for lane in (5, 97):
        iv = [int.from_bytes(inst[lane, k].tobytes(), 'little') for k in range(wl.n_instance)]
        wv = [int.from_bytes(wit[lane, k].tobytes(), 'little') for k in range(wl.n_witness)]
        ref = oracle_lane(wl.mod_le, iv, wv, msgs, wl.width, trace=False)
        assert ev.get_violations(lane) == ref.violations


def test_full_size_c2_outputs_match_the_committed_oracle_digests():
    """tests/golden/c2_digests.json: SHA-256 of the 64 output wires of six lanes of the BASELINE configs[1] relation as
    the oracle computes them (fixture committed with its generator).  The production schedule (fusion, pair entries,
    XCD-aware grid) must reproduce them bit for bit."""
    import hashlib
    import json
    import os
    from helpers import ROOT
    fx = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'c2_digests.json')))
    wl = workloads.ArithLayered()
    batch = 1024
    inst, wit = wl.inputs(batch)
    ev = zk.Evaluator()
    ev.declare_inputs(wl.n_instance0, wl.n_witness)
    for m in wl.relation_messages(with_epilogue=False, free_last=False):
        ev.ingest_message(m)
    ev.finalize()
    assert ev.schedule_info()['device_ops'] < 700000   # the compact program, not the retain_all one
    ev.set_inputs(np.ascontiguousarray(inst[:, :wl.n_instance0]).tobytes(), wit.tobytes(), batch)
    ev.replay()
    ev.synchronize()
    cols = [ev.get(w, batch) for w in wl.output_wire_ids()]
    for lane, want in fx['lanes'].items():
        vals = [col[int(lane)] for col in cols]
        assert hashlib.sha256('\n'.join(str(v) for v in vals).encode()).hexdigest() == want['sha256'], lane
        assert str(vals[0]) == want['first_output']
    # ... and ALL 1024 lanes against the per-lane hashes of tests/golden/c2_all_lanes.json (cpu_opt, pinned to the oracle
    # on the six lanes above by tests/test_oracle_golden.py)
    import cpu_checkers
    outs = np.zeros((batch, wl.n_out, wl.width), dtype=np.uint8)
    for t, col in enumerate(cols):
        outs[:, t] = np.frombuffer(b''.join(v.to_bytes(wl.width, 'little') for v in col), dtype=np.uint8).reshape(batch, wl.width)
    cpu_checkers.check_against_all_lanes_fixture(json.load(open(os.path.join(ROOT, 'tests', 'golden', 'c2_all_lanes.json'))), outs)


@pytest.mark.parametrize('batch', [1100, 2560])
def test_grid_mappings_and_op_orders_agree(batch):
    """Every launch geometry gives the same per-lane answer as the CPU evaluator of the same tape: XCD-aware grid
    (2560 lanes = 40 lane blocks: shares of 16 + 24, two and three groups per XCD) and the plain grid (1100
    lanes = 18 lane blocks, not a multiple of 8), two ops per wave on 4096-wide levels, the three op orders,
    one to three streams, explicit lane groups."""
    from oracle_lib import opt_eval
    wl = workloads.ArithLayered(W=4096, D=5, n_instance0=64, n_out=16)
    ev, inst, wit, n_bad = _layered_session(wl, batch)
    ev.replay()
    ev.synchronize()
    base = ev.lane_results(batch)
    assert ev.counts() == (batch - n_bad, n_bad)
    kinds, ta, tb = ev.tape()
    ff_cpu, _, _ = opt_eval(kinds, ta, tb, ev.constants(), wl.mod_le, inst.tobytes(), wl.n_instance, wit.tobytes(),
                            wl.n_witness, wl.width, batch, 8)
    assert np.array_equal(ff_cpu, base[0])
    for opts in ({'xcd_map': 0}, {'streams': 1}, {'streams': 3}, {'xcd_map': 1, 'streams': 2}):
        for k, v in opts.items():
            ev.set_option(k, str(v))
        for group in (0, 512, 1024):
            ev.set_lane_group(group)
            ev.replay()
            ev.synchronize()
            got = ev.lane_results(batch)
            assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1]), (opts, group)
    for order in (0, 1):
        other = zk.Evaluator()
        other.set_option('sort_by_operand', str(order))
        other.declare_inputs(wl.n_instance, wl.n_witness)
        for m in wl.relation_messages():
            other.ingest_message(m)
        other.finalize()
        other.set_inputs(inst.tobytes(), wit.tobytes(), batch)
        other.replay()
        other.synchronize()
        assert np.array_equal(other.lane_results(batch)[0], base[0]), order


@pytest.mark.parametrize('seed', range(25, 45))
def test_random_structured_relations_on_gpu_compact_schedule(seed):
    """same fuzz through the production schedule (slot reuse, operand ordering, gate fusion, two
    streams): violation strings, counts and the surviving top-level wires against the oracle."""
    from random_circuits import Gen
    from test_fuzz_host import FIELDS
    p, boolean = FIELDS[seed % len(FIELDS)]
    g = Gen(seed, p, boolean)
    rel, mod_le = g.relation(n_top=14)
    lanes = 130
    rows_i, rows_w = g.lane_inputs(lanes, seed + 1000)
    ev = zk.Evaluator()
    ev.declare_inputs(g.n_inst, g.n_wit)
    ev.ingest_message(rel)
    ev.finalize(retain_all=False)
    inst, wit = batch_arrays(rows_i, rows_w, ev.elem_bytes)
    ev.set_inputs(inst if g.n_inst else None, wit if g.n_wit else None, lanes)
    ev.replay()
    ev.synchronize()
    n_ok = 0
    for lane in range(0, lanes, 3):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32, trace=False)
        assert ev.get_violations(lane) == ref.violations, (seed, lane)
    first, _ = ev.lane_results(lanes)
    for lane in range(lanes):
        n_ok += int(first[lane]) == zk.NO_FAIL
    assert ev.counts() == (n_ok, lanes - n_ok)
    # Evaluator::get on wires still alive at the end (pinned: never fused away, never overwritten)
    ref = oracle_lane(mod_le, rows_i[0], rows_w[0], [rel], 32, trace=False)
    if not ref.violations:
        checked = 0
        for wid in range(0, 60):
            want = ref.get(wid)
            got = ev.get(wid, lanes)
            assert (want is None) == (got is None), (seed, wid)
            if want is not None:
                assert got[0] == want, (seed, wid)
                checked += 1
        assert checked > 0


@pytest.mark.parametrize('seed', range(25))
def test_random_structured_relations_on_gpu(seed):
    """functions / for / switch / nested switch / frees: every lane's wire values and violation
    strings against the oracle (same generator as the CPU-tier fuzz)."""
    from random_circuits import Gen
    from test_fuzz_host import FIELDS
    p, boolean = FIELDS[seed % len(FIELDS)]
    g = Gen(seed, p, boolean)
    rel, mod_le = g.relation(n_top=14)
    lanes = 67
    rows_i, rows_w = g.lane_inputs(lanes, seed + 1000)
    ev = zk.Evaluator()
    ev.declare_inputs(g.n_inst, g.n_wit)
    ev.ingest_message(rel)
    ev.finalize(retain_all=True)
    w = ev.elem_bytes
    inst, wit = batch_arrays(rows_i, rows_w, w)
    ev.set_inputs(inst if g.n_inst else None, wit if g.n_wit else None, lanes)
    ev.replay()
    ev.synchronize()
    vals = ev.dump_trace_values(lanes)
    n_ok = 0
    for lane in range(lanes):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32)
        assert ev.get_violations(lane) == ref.violations, (seed, lane)
        rv = ref.trace_values()
        assert vals[lane][:len(rv)] == rv, (seed, lane)
        n_ok += not ref.violations
    assert ev.counts() == (n_ok, lanes - n_ok)


@pytest.mark.parametrize('path', ['hbm', 'lds'])
@pytest.mark.parametrize('name', ['bool_correct', 'bool_incorrect'])
def test_boolean_paths_agree_with_oracle(name, path):
    """GF(2): the HBM-table kernel and the LDS-resident kernel, both against the oracle."""
    bufs = golden_buffers(name)
    for retain in (True, False):
        ev = zk.Evaluator.from_messages(bufs)
        ev.set_option('bool_path', path)
        ev.finalize(retain_all=retain)
        ev.set_inputs_from_messages()
        assert ev.uses_lds_path() == (path == 'lds')
        ev.replay()
        ev.synchronize()
        ref = OracleRun(buffers=bufs)
        assert ev.get_violations(0) == ref.violations
        if retain:
            rv = ref.trace_values()
            assert ev.dump_trace_values(1)[0][:len(rv)] == rv


@pytest.mark.parametrize('path', ['hbm', 'lds'])
def test_boolean_layered_batch_against_oracle(path):
    """small C4-shaped relation, ragged batch spanning several 32- and 64-witness words"""
    wl = workloads.BoolLayered(W=128, D=10, n_instance0=16, n_out=8)
    batch = 203
    inst, wit = wl.inputs(batch)
    probe = zk.Evaluator()
    probe.set_option('bool_path', path)
    probe.declare_inputs(wl.n_instance0, wl.n_witness)
    for m in wl.relation_messages(with_epilogue=False, free_last=False):
        probe.ingest_message(m)
    probe.finalize()
    probe.set_inputs(np.ascontiguousarray(inst[:, :wl.n_instance0]).tobytes(), wit.tobytes(), batch)
    probe.replay()
    probe.synchronize()
    outs = np.zeros((batch, wl.n_out), dtype=np.uint8)
    for t, wid in enumerate(wl.output_wire_ids()):
        outs[:, t] = probe.get(wid, batch)   # Evaluator::get on live wires (LDS path: write-back)
    lane = 77
    ref = oracle_lane(wl.mod_le, inst[lane, :wl.n_instance0, 0].tolist(), wit[lane, :, 0].tolist(),
                      wl.relation_messages(with_epilogue=False, free_last=False), 1, trace=False)
    assert [ref.get(w) for w in wl.output_wire_ids()] == outs[lane].tolist()
    n_bad = wl.set_expected_outputs(inst, outs)
    ev = zk.Evaluator()
    ev.set_option('bool_path', path)
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    msgs = wl.relation_messages()
    for m in msgs:
        ev.ingest_message(m)
    ev.finalize()
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (batch - n_bad, n_bad)
    for lane in (0, 31, 32, 63, 64, 97, 194, 202):
        ref = oracle_lane(wl.mod_le, inst[lane, :, 0].tolist(), wit[lane, :, 0].tolist(), msgs, 1, trace=False)
        assert ev.get_violations(lane) == ref.violations, lane
    # a value > 1 is not a canonical GF(2) element: its low bit is all and / xor ever look at (the reference's
    # `(a & b) % 2`), but `not` tests the integer for zero -- a lane is refused only if the value can reach a `not`
    inst2 = inst.copy()
    inst2[5, 0, 0] = 3
    ev.set_inputs(inst2.tobytes(), wit.tobytes(), batch)
    ev.replay()
    ev.synchronize()
    got = ev.get_violations(5)
    ref = oracle_lane(wl.mod_le, inst2[5, :, 0].tolist(), wit[5, :, 0].tolist(), msgs, 1, trace=False)
    assert got == ref.violations or (len(got) == 1 and 'not canonical' in got[0])


@pytest.mark.parametrize('name', ['arith_101_correct', 'arith_101_incorrect', 'arith_bn254_correct'])
def test_r1cs_of_a_relation_is_satisfied_by_the_replayed_wires(name):
    """ir-to-zkif: rows from the tape (to_r1cs.rs rules), assignment = the wire table, row check on the
    GPU.  A statement is TRUE iff its R1CS instance is satisfied; a failing assert is its own row."""
    bufs = golden_buffers(name)
    ev = zk.Evaluator.from_messages(bufs)
    ev.finalize(retain_all=True)
    ev.r1cs_from_tape()
    ev.set_inputs_from_messages()
    ev.replay()
    ev.r1cs_check()
    ff, counts = ev.r1cs_results(1)
    ref = OracleRun(buffers=bufs)
    assert counts == ((1, 0) if not ref.violations else (0, 1))
    rows, _ = ev.r1cs_export()
    if ref.violations:
        assert rows[int(ff[0])][2] == [(0, 0)]  # the violated row is an assert_zero row
    else:
        assert int(ff[0]) == zk.NO_FAIL


@pytest.mark.parametrize('p', [101, circuits.BN254_R, 2 ** 61 - 1])
def test_r1cs_quotient_wires_on_the_gpu(p):
    """ToR1CSConverter with use_correction gives every add / mul / add_constant / mul_constant call a second variable,
    the integer quotient q = (a op b) / p (to_r1cs.rs:163-211,213-260,262-359), so that `a op b = out + q * p` holds over
    the integers.  The GPU computes it for a whole batch from the retain_all wire table; here against Python integers on
    the oracle's wire values of every lane, for all such calls of the example relation (ladders included)."""
    lanes = 5
    _, _, rel = circuits.arith_example(p)
    rows_i, rows_w = _batched_example(p, lanes)
    ev = zk.Evaluator()
    ev.declare_inputs(3, 4)
    ev.ingest_message(rel)
    ev.finalize(retain_all=True)
    ev.r1cs_from_tape(use_correction=True)
    inst, wit = batch_arrays(rows_i, rows_w, ev.elem_bytes)
    ev.set_inputs(inst, wit, lanes)
    ev.replay()
    ev.synchronize()
    kinds, a, b = ev.tape()
    consts = [int.from_bytes(c, 'little') for c in ev.constants()]
    calls = [i for i, k in enumerate(kinds) if int(k) in (1, 2, 3, 4)]
    assert len(calls) > 50
    got = ev.r1cs_correction_values(calls, lanes)
    value_index = {i: t for t, i in enumerate(j for j, k in enumerate(kinds) if int(k) != 9)}
    mod_le = p.to_bytes((p.bit_length() + 7) // 8, 'little')
    for lane in range(lanes):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32)
        vals = ref.trace_values()
        checked = 0
        for n, i in enumerate(calls):
            if value_index[i] >= len(vals):
                break                      # the oracle stopped at this lane's first failing assert
            x = vals[value_index[int(a[i])]]
            k = int(kinds[i])
            y = vals[value_index[int(b[i])]] if k in (1, 2) else consts[int(b[i])]
            full = x + y if k in (1, 3) else x * y
            assert full % p == vals[value_index[i]]
            assert got[lane][n] == full // p, (lane, i, k)
            checked += 1
        assert checked > 30
    with pytest.raises(zk.ZkGpuError, match='is not add / mul'):
        ev.r1cs_correction_values([i for i, k in enumerate(kinds) if int(k) == 5][:1], lanes)


def test_r1cs_batched_matches_evaluation_counts():
    lanes = 70
    _, _, rel = circuits.arith_example(circuits.BN254_R)
    rows_i, rows_w = _batched_example(circuits.BN254_R, lanes)
    ev = zk.Evaluator()
    ev.declare_inputs(3, 4)
    ev.ingest_message(rel)
    ev.finalize(retain_all=True)
    ev.r1cs_from_tape()
    inst, wit = batch_arrays(rows_i, rows_w, ev.elem_bytes)
    ev.set_inputs(inst, wit, lanes)
    ev.replay()
    ev.synchronize()
    ev.r1cs_check()
    ff, counts = ev.r1cs_results(lanes)
    assert counts == ev.counts()
    first, _ = ev.lane_results(lanes)
    assert [(int(x) == zk.NO_FAIL) for x in ff] == [(int(x) == zk.NO_FAIL) for x in first]


SECP256K1_P = 2 ** 256 - 2 ** 32 - 977   # top bit set: twice a canonical value does not fit the eight words


@pytest.mark.parametrize('p,coef_kind,classes', [
    (101, 'random', True), (circuits.BN254_R, 'random', True),
    # coefficients 1 / -1 / small signed integers: the unit and small coefficient classes of the row kernel
    # (device/args.hpp kR1csClass*), against the same Python integers; and the same rows with the classes turned off
    (101, 'small', True), (circuits.BN254_R, 'small', True), (circuits.BN254_R, 'small', False), (2 ** 61 - 1, 'small', True),
    (SECP256K1_P, 'small', True), (circuits.P512, 'small', True)])
def test_r1cs_csr_rows_with_coefficients(p, coef_kind, classes):
    """caller-supplied CSR (3+3 term products with random coefficients): witness generation by the
    row kernel level by level, values against Python integers, then the check; one false row."""
    wl = workloads.R1csSynthetic(M=300, n_base=24, n_coefs=50, seed=5, p=p, coef_kind=coef_kind)
    batch = 67
    ev = zk.Evaluator()
    ev.set_option('r1cs_coef_classes', '1' if classes else '0')
    ev.declare_inputs(0, wl.n_witness)
    ev.ingest_message(wl.base_relation())
    ev.finalize(retain_all=True)
    row_ptr, tv, tc, cb = wl.csr()
    # append a false row: (z_0) * (one) = (z_0 + 1)
    one = len(cb) - 1
    z0 = wl.n_base + 1
    t0 = int(row_ptr[-1])
    row_ptr = np.concatenate([row_ptr, np.array([t0 + 1, t0 + 2, t0 + 4], dtype=np.uint32)])
    tv = np.concatenate([tv, np.array([z0, 2 ** 64 - 1, z0, 2 ** 64 - 1], dtype=np.uint64)])
    tc = np.concatenate([tc, np.array([one] * 4, dtype=np.uint32)])
    ev.r1cs_load_csr(row_ptr, tv, tc, cb, wl.width, wl.M)
    w = wl.witnesses(batch)
    ev.set_inputs(None, w.tobytes(), batch)
    ev.replay()
    lo = 0
    for hi in wl.level_bounds:
        ev.r1cs_assign(lo, int(hi) - lo)
        lo = int(hi)
    assert lo == wl.M
    # expected output variable E := z_last (all lanes honest), reload the base variables
    zl = ev.r1cs_get_var(wl.last_z, batch)
    for lane in range(batch):
        w[lane, wl.n_base] = np.frombuffer(zl[lane].to_bytes(wl.width, 'little'), dtype=np.uint8)
    ev.set_inputs(None, w.tobytes(), batch)
    ev.replay()
    # values of a few z against Python integers
    coefs = [int.from_bytes(cb[i].tobytes(), 'little') for i in range(len(cb))]
    for lane in (0, 33, batch - 1):
        val = {k: int.from_bytes(w[lane, k].tobytes(), 'little') for k in range(wl.n_witness)}
        for r in range(wl.M):
            terms = [(int(tv[7 * r + k]), coefs[int(tc[7 * r + k])]) for k in range(6)]
            a = sum(c * val[v] for v, c in terms[:3]) % p
            b = sum(c * val[v] for v, c in terms[3:]) % p
            val[wl.n_base + 1 + r] = a * b % p
        for var in (wl.n_base + 1, wl.n_base + 1 + wl.M // 2, wl.last_z):
            assert ev.r1cs_get_var(var, batch)[lane] == val[var]
    ev.r1cs_check()
    ff, counts = ev.r1cs_results(batch)
    assert counts == (0, batch)            # the appended false row fails in every lane
    assert all(int(x) == wl.M + 1 for x in ff)
    # the CPU row check (oracle/cpu_opt.cpp, the cpu_baseline of bench.py --workload c5) sees the same first failing row
    if wl.width <= 32:     # (that checker is written for four 64-bit limbs)
        from oracle_lib import r1cs_check
        ff_cpu, _ = r1cs_check(row_ptr, tv, tc, cb, wl.mod_le, w, wl.n_base + 1 + wl.M, wl.M, 4)
        assert np.array_equal(ff_cpu, ff)
    cc = ev.r1cs_class_counts()
    assert (cc['unit'] > 0 and cc['small'] > 0) == (coef_kind == 'small' and classes), cc


def test_synth_workspace_round_trips_through_files_and_oracle(tmp_path):
    """`cli.py synth c2`: a benchmark statement written as a real workspace evaluates TRUE through the
    file Source of both the product and the oracle; the --incorrect variant is FALSE with the same text."""
    import io
    from zkinterface_ir_amd import cli
    for corrupt in (False, True):
        d = str(tmp_path / ('ws_bad' if corrupt else 'ws'))
        assert cli.synth('c2', d, lane=3, corrupt=corrupt, err=io.StringIO()) == 0
        ref = OracleRun(files=[d + '/002_relation.sieve', d + '/000_instance.sieve', d + '/001_witness.sieve'], trace=False)
        assert zk.evaluate([d]) == ref.violations
        assert (ref.violations == []) == (not corrupt)


def test_trait_level_recording_replays_on_gpu():
    """a tape recorded call by call through zkgpu_backend_* (what the Rust `impl ZKBackend` does)"""
    from test_host_tape import _record_through_trait
    p = circuits.BN254_R
    ev = zk.Evaluator()
    handles = _record_through_trait(ev, p)
    ev.finalize(retain_all=True)
    lanes = [(5, 9, 28), (5, 9, 29), (0, 0, 3), (1, 0, 5), (p - 1, 1, 4)]
    inst, wit = batch_arrays([[i0] for _, _, i0 in lanes], [[w0, w1] for w0, w1, _ in lanes], ev.elem_bytes)
    ev.set_inputs(inst, wit, len(lanes))
    ev.replay()
    ev.synchronize()
    vals = ev.dump_trace_values(len(lanes))
    for l, (w0, w1, i0) in enumerate(lanes):
        z = (w0 * w0 + 3 - i0) % p
        assert vals[l][6] == z and vals[l][9] == (w1 * 7) % p and vals[l][10] == (w1 * 7 * z) % p
        expect = [] if z == 0 else ['Wire_41 (may be weighted) should be 0, while it is not']
        assert ev.get_violations(l) == expect
    assert ev.counts() == (3, 2)


@pytest.mark.parametrize('shrink', [False, True])
def test_trait_level_recording_with_a_field_change_replays_on_gpu(shrink):
    """zkgpu_backend_set_field with another modulus (what the reference's Evaluator does for every Relation message,
    evaluator.rs:262-268) opens a field segment; the handles that were not dropped live on.  Every lane's violations
    against the oracle's run of the equivalent two-message relation."""
    import test_field_segments as fs
    from zkinterface_ir_amd import sieve_writer as sw
    parts, p2 = (fs.SHRINK, fs.P3) if shrink else (fs.GROW, fs.P2)
    msgs = fs._messages(parts)
    ev = fs._trait_session(p2, shrink)
    ev.finalize()
    assert ev.n_field_segments == 2
    rows = [([e0], [w0, w1]) for w0, w1, e0 in ((0, 5, 0), (97, 1, 0), (97, 1, 97), (100, 1, 3), (100, 1, 100), (98, 99, (98 * 99) % 101))] \
        if shrink else fs._lanes_grow()
    inst, wit = batch_arrays([r[0] for r in rows], [r[1] for r in rows], ev.elem_bytes)
    ev.set_inputs(inst, wit, len(rows))
    ev.replay()
    ev.synchronize()
    n_ok = 0
    for lane, (i, w) in enumerate(rows):
        ref = oracle_lane(sw.int_to_le(fs.P1), i, w, msgs, 32, trace=False)
        assert ev.get_violations(lane) == ref.violations, (lane, i, w)
        n_ok += ref.violations == []
    assert ev.counts() == (n_ok, len(rows) - n_ok)


@pytest.mark.parametrize('name', ['with_function', 'with_several_functions', 'switch_builder', 'switch_nested_in_function'])
def test_builder_circuits_replay_true(name):
    """rust/src/producers/builder.rs:727-1175: the four GateBuilder circuits evaluate with zero violations."""
    from builder_circuits import BUILDER_CIRCUITS
    bufs = BUILDER_CIRCUITS[name]().buffers()
    assert zk.evaluate(bufs) == []
    ref = OracleRun(buffers=bufs)
    ev = zk.Evaluator.from_messages(bufs)
    ev.finalize(retain_all=True)
    ev.set_inputs_from_messages()
    ev.replay()
    ev.synchronize()
    assert ev.dump_trace_values(1)[0] == ref.trace_values()


def test_r1cs_example_converted_to_ir_replays_with_reference_wire_values(tmp_path):
    """rust/src/producers/from_r1cs.rs:176-217: wires 0..6 hold 1, 100, 3, 4, 25, 9, 16 and nothing is violated;
    through a FilesSink workspace as `zkif-to-ir` would write it (cli.rs:365-440)."""
    from builder_circuits import R1CS_EXAMPLE_WIRES, r1cs_example
    from zkinterface_ir_amd.builder import FilesSink
    sink = r1cs_example(FilesSink.new_clean(str(tmp_path / 'ws')))
    sink.close()
    ev = zk.Evaluator()
    ev.ingest_paths(sink.paths())
    ev.finalize(retain_all=True)
    ev.set_inputs_from_messages()
    ev.replay()
    ev.synchronize()
    assert ev.get_violations(0) == []
    assert [ev.get(w, 1)[0] for w in range(7)] == R1CS_EXAMPLE_WIRES
    assert zk.evaluate(r1cs_example(zz=26).buffers()) == ['Wire_35 (may be weighted) should be 0, while it is not']


@pytest.mark.parametrize('p', [101, circuits.BN254_R])
def test_r1cs_rows_and_their_ir_expansion_agree_per_lane(p):
    """One constraint system, two device paths: the row-check kernel over the CSR (C5) and the replay of the
    FromR1CSConverter expansion of the same rows (C2 path).  Every lane must get the same verdict, and the first
    failing row must be the first failing assert."""
    from zkinterface_ir_amd.builder import MemorySink
    from zkinterface_ir_amd.from_r1cs import FromR1CSConverter
    rng = np.random.default_rng(1234)
    n_base, M, batch = 12, 120, 70
    width = 8 * ((p.bit_length() + 63) // 64)
    le = lambda x: int(x).to_bytes(width, 'little')
    rows = []   # (A, B, C) with terms (var id, coef int); ids: 0 = one, 1..n_base base, n_base+1+i = z_i
    for i in range(M):
        hi = n_base + 1 + i
        lc = lambda k: [(int(rng.integers(0, hi)), int(rng.integers(0, 2 ** 62)) % p) for _ in range(k)]
        rows.append((lc(int(rng.integers(1, 4))), lc(int(rng.integers(0, 4))), [(hi, 1)]))
    # per-lane assignments; some lanes get one wrong z
    vals = np.zeros((batch, n_base + 1 + M), dtype=object)
    bad_row = {}
    for lane in range(batch):
        vals[lane, 0] = 1
        for k in range(1, n_base + 1):
            vals[lane, k] = int(rng.integers(0, 2 ** 62)) % p
        wrong = int(rng.integers(0, M)) if lane % 5 == 2 else None
        for i, (a, b, _c) in enumerate(rows):
            z = sum(c * vals[lane, v] for v, c in a) % p * (sum(c * vals[lane, v] for v, c in b) % p) % p
            if wrong == i:
                z = (z + 1) % p
                bad_row[lane] = i
            vals[lane, n_base + 1 + i] = z
    n_wit = n_base + M
    wit = np.frombuffer(b''.join(le(vals[lane, k]) for lane in range(batch) for k in range(1, n_wit + 1)),
                        dtype=np.uint8).reshape(batch, n_wit, width)

    # path 1: CSR rows through the R1CS kernel; variable id k (>= 1) is witness k - 1, id 0 is the constant one
    cb = [le(1)]
    tv, tc, row_ptr = [], [], [0]
    for a, b, c in rows:
        for lcomb in (a, b, c):
            for var, coef in lcomb:
                tv.append(2 ** 64 - 1 if var == 0 else var - 1)
                cb.append(le(coef))
                tc.append(len(cb) - 1)
            row_ptr.append(len(tv))
    ev = zk.Evaluator()
    ev.declare_inputs(0, n_wit)
    ev.ingest_message(sw.write_relation(sw.int_to_le(p), 'arithmetic', 'simple', [], [('witness', k) for k in range(n_wit)]))
    ev.finalize(retain_all=True)
    ev.r1cs_load_csr(np.array(row_ptr, dtype=np.uint32), np.array(tv, dtype=np.uint64), np.array(tc, dtype=np.uint32),
                     np.frombuffer(b''.join(cb), dtype=np.uint8).reshape(len(cb), width), width, 0)
    ev.set_inputs(None, wit.tobytes(), batch)
    ev.replay()
    ev.r1cs_check()
    ff_rows, counts_rows = ev.r1cs_results(batch)

    # path 2: the same rows expanded to gates, replayed for the same lanes
    conv = FromR1CSConverter(MemorySink(), p - 1, [(0, le(1))], list(range(1, n_wit + 1)))
    conv.ingest_constraints([tuple([(var, le(c)) for var, c in lcomb] for lcomb in row) for row in rows])
    rel = conv.finish().buffers()[2]
    ev2 = zk.Evaluator()
    ev2.declare_inputs(0, n_wit)
    ev2.ingest_message(rel)
    ev2.finalize()
    ev2.set_inputs(None, wit.tobytes(), batch)
    ev2.replay()
    ev2.synchronize()
    ff_gates, flags = ev2.lane_results(batch)
    assert ev2.counts() == counts_rows == (batch - len(bad_row), len(bad_row))
    for lane in range(batch):
        if lane in bad_row:
            assert int(ff_rows[lane]) == bad_row[lane] and int(ff_gates[lane]) == bad_row[lane]
        else:
            assert int(ff_rows[lane]) == 0xFFFFFFFF and int(ff_gates[lane]) == 0xFFFFFFFF


def test_graph_replay_gives_the_same_results():
    """The captured-hipGraph replay path (off by default: slower on ROCm 7.2) and the stream path agree, also after the
    inputs, the batch size or an option change invalidate the capture."""
    _, _, rel = circuits.arith_example(circuits.BN254_R)
    ev = zk.Evaluator()
    ev.declare_inputs(3, 4)
    ev.ingest_message(rel)
    ev.finalize()
    for lanes in (70, 200):
        rows_i, rows_w = _batched_example(circuits.BN254_R, lanes)
        inst, wit = batch_arrays(rows_i, rows_w, 32)
        results = {}
        for mode in ('0', '1', '1', '0'):
            ev.set_option('graph', mode)
            ev.set_inputs(inst, wit, lanes)
            ev.replay()
            ev.replay()          # a second launch of the same capture
            ev.synchronize()
            results.setdefault(mode, []).append((ev.counts(), ev.lane_results(lanes)[0].tolist()))
        ev.set_option('streams', '1')
        ev.set_option('graph', '1')
        ev.replay()
        ev.synchronize()
        results['1'].append((ev.counts(), ev.lane_results(lanes)[0].tolist()))
        ev.set_option('streams', '2')
        flat = results['0'] + results['1']
        assert all(r == flat[0] for r in flat)
        assert 0 < flat[0][0][0] < lanes


def test_inputs_handed_over_while_the_previous_batch_replays():
    """zkgpu_set_inputs fills the input set the replay in flight is not reading (copy stream, two sets): a stream of
    different batches, each handed over right after the previous replay was queued, gives every batch its own answer."""
    wl = workloads.ArithLayered(W=512, D=6, n_instance0=16, n_out=8)
    batch = 256
    ev, inst, wit, n_bad = _layered_session(wl, batch)
    good = (inst.copy(), wit.copy())
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (batch - n_bad, n_bad)
    bad_w = wit.copy()
    bad_w[:, :, 0] ^= 1                       # every witness value of every lane changed: every statement false
    answers = {True: (batch - n_bad, n_bad), False: (0, batch)}
    pending = None                            # answer the replay in flight must give
    for k in range(8):
        is_good = k % 2 == 0
        w = good[1] if is_good else bad_w
        ev.set_inputs(inst.tobytes(), w.tobytes(), batch)   # handed over while replay k-1 is still queued or running
        if pending is not None:
            assert ev.counts() == pending, 'replay %d read the inputs handed over for replay %d' % (k - 1, k)
        ev.replay()
        pending = answers[is_good]
    assert ev.counts() == pending             # the last batch handed over was the damaged one
    assert pending == (0, batch)
    ev.set_inputs(inst.tobytes(), good[1].tobytes(), batch)
    ev.set_inputs(inst.tobytes(), bad_w.tobytes(), batch)   # two uploads in a row, then the good one again
    ev.set_inputs(inst.tobytes(), good[1].tobytes(), batch)
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (batch - n_bad, n_bad)
    # page-locked caller memory is read by the DMA engine directly (no staging copy): same answers.  The buffers come
    # from the HIP runtime the library itself links (a torch pin_memory() here would bring up torch's own bundled
    # runtime after the library's, which a fresh box does not always survive)
    import ctypes
    hip = ctypes.CDLL('libamdhip64.so')

    def pinned(arr):
        arr = np.ascontiguousarray(arr).reshape(-1)
        p = ctypes.c_void_p()
        assert hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(arr.nbytes), ctypes.c_uint(0)) == 0
        ctypes.memmove(p, arr.ctypes.data, arr.nbytes)
        return p

    pin_i, pin_bad, pin_good = pinned(inst), pinned(bad_w), pinned(good[1])
    try:
        for pin_w, want in ((pin_bad, (0, batch)), (pin_good, (batch - n_bad, n_bad)), (pin_bad, (0, batch))):
            ev.set_inputs(pin_i.value, pin_w.value, batch)
            ev.replay()
            ev.synchronize()
            assert ev.counts() == want
    finally:
        ev.synchronize()
        for p in (pin_i, pin_bad, pin_good):
            hip.hipHostFree(p)


@pytest.mark.gpu
@pytest.mark.parametrize('p', [circuits.BN254_R, 2 ** 61 - 1, SECP256K1_P, 101])
def test_r1cs_small_coefficient_class_at_its_bounds(p):
    """the small-coefficient path of the row kernel sums |c| * v as an integer of N + 2 words: 255 terms of magnitude
    2^31 - 1 on values p - 1 is the largest it can meet (device/r1cs_kernels.hpp r1cs_lincomb_small); all signs; products
    of two such sums; a full-class combination beside a small one (the scales of the two sides of a check differ);
    assignment (the scale undone by a product) and check, against Python integers"""
    import numpy as np
    rng = np.random.default_rng(7)
    width = 8 * ((p.bit_length() + 63) // 64)
    n_base, batch = 260, 70
    big = (2 ** 31 - 1) % p or 1
    pool = [1, big, (p - big) % p or 1, p - 1, 2, (p // 3) | 1 if p > 2 ** 40 else 5]
    ONE, BIG, MBIG, MONE, TWO, WIDE = range(6)
    cb = np.frombuffer(b''.join(v.to_bytes(width, 'little') for v in pool), dtype=np.uint8).reshape(len(pool), width)
    Z = n_base          # extra variables z0.. start here
    rows = [
        ([(k, BIG) for k in range(255)], [(2 ** 64 - 1, ONE)], [(Z + 0, ONE)]),                          # z0 = sum big * v
        ([(k, MBIG) for k in range(255)], [(k, BIG) for k in range(200)], [(Z + 1, ONE)]),                # z1 = (-sum) * (sum)
        ([(k, MONE if k % 2 else ONE) for k in range(40)], [(7, TWO), (8, MBIG)], [(Z + 2, ONE)]),        # unit x small
        ([(3, WIDE), (4, BIG)], [(5, MBIG), (6, TWO), (9, ONE)], [(Z + 3, ONE)]),                         # full x small
    ]
    n_assign = len(rows)
    # rows that only the check sees: true by construction, and one that is false wherever variable 11 is not 0
    rows += [
        ([(k, BIG) for k in range(255)], [(2 ** 64 - 1, ONE)], [(k, BIG) for k in range(255)]),          # small = small
        ([(Z + 0, ONE)], [(2 ** 64 - 1, ONE)], [(k, BIG) for k in range(255)]),                          # full (one variable) = small
        ([(k, MBIG) for k in range(255)], [(k, BIG) for k in range(200)], [(Z + 1, ONE)]),               # small * small = full
        ([(3, WIDE), (4, BIG)], [(5, MBIG), (6, TWO), (9, ONE)], [(Z + 3, TWO), (Z + 3, MONE)]),         # full * small = small (2z - z)
        ([(10, TWO), (11, BIG)], [(2 ** 64 - 1, ONE)], [(10, TWO), (11, MBIG)]),                         # false unless v11 = 0
    ]
    starts, tv, tc = [], [], []
    for parts in rows:
        for part in parts:
            starts.append(len(tv))
            tv += [v for v, _ in part]
            tc += [c for _, c in part]
    starts.append(len(tv))
    ev = zk.Evaluator()
    ev.declare_inputs(0, n_base)
    from zkinterface_ir_amd.sieve_writer import write_relation
    ev.ingest_message(write_relation(p.to_bytes((p.bit_length() + 7) // 8, 'little'), 'arithmetic', 'simple', [],
                                     [('witness', k) for k in range(n_base)]))
    ev.finalize(retain_all=True)
    ev.r1cs_load_csr(np.array(starts, dtype=np.uint32), np.array(tv, dtype=np.uint64), np.array(tc, dtype=np.uint32), cb, width, n_assign)
    cc = ev.r1cs_class_counts()
    assert cc['small'] >= 9 and cc['unit'] >= 1, cc
    vals = [[int(rng.integers(0, 2 ** 62)) ** 5 % p for _ in range(n_base)] for _ in range(batch)]
    for lane in range(batch):
        if lane % 3 == 0:
            vals[lane] = [p - 1] * n_base          # the largest sums
        if lane % 5 == 0:
            vals[lane][11] = 0                     # the false row holds here
    w = np.zeros((batch, n_base, width), dtype=np.uint8)
    for lane in range(batch):
        for k in range(n_base):
            w[lane, k] = np.frombuffer(vals[lane][k].to_bytes(width, 'little'), dtype=np.uint8)
    ev.set_inputs(None, w.tobytes(), batch)
    ev.replay()
    for r in range(n_assign):          # one row per call: z0 is read by no later assigned row, but keep the order
        ev.r1cs_assign(r, 1)

    def comb(part, v):
        return sum(pool[c] * (1 if var == 2 ** 64 - 1 else v[var]) for var, c in part) % p
    got = ev.r1cs_get_vars([Z + k for k in range(n_assign)], batch)
    for lane in range(batch):
        v = list(vals[lane])
        for r in range(n_assign):
            v.append(comb(rows[r][0], v) * comb(rows[r][1], v) % p)
        assert got[lane] == v[n_base:], lane
    ev.r1cs_check()
    ff, counts = ev.r1cs_results(batch)
    false_row = len(rows) - 1
    # the last row: 2 v10 + big v11 = 2 v10 - big v11  <=>  2 big v11 = 0  <=>  v11 = 0 (p odd, big != 0 mod p ... or p | 2 big)
    for lane in range(batch):
        holds = (2 * pool[BIG] * vals[lane][11]) % p == 0
        assert (int(ff[lane]) == zk.NO_FAIL) == holds and (holds or int(ff[lane]) == false_row), (lane, int(ff[lane]))
