"""CPU tier: the oracle against every golden vector the reference's own tests hold
for the evaluate path, and the trace digests of SURVEY.md Appendix A."""
from collections import Counter

import pytest

import circuits
from helpers import REF_EXAMPLES, golden_buffers
from oracle_lib import OracleRun, oracle_exp


@pytest.mark.parametrize('name', sorted(circuits.GOLDEN_TRACES))
def test_trace_digest_and_verdict(name):
    n_ops, sha, violations = circuits.GOLDEN_TRACES[name]
    run = OracleRun(buffers=golden_buffers(name))
    assert run.violations == violations
    assert len(run.trace_kinds()) == n_ops
    assert run.trace_sha256() == sha


def test_committed_fixture_via_file_source():
    # rust/examples/*.sieve read through the Source ordering rule (source.rs:69-89):
    # the paths are handed over in the wrong order on purpose.
    run = OracleRun(files=list(reversed(REF_EXAMPLES)))
    assert run.violations == []
    assert Counter(run.trace_kinds()) == {'copy': 99, 'add': 43, 'mul': 37, 'constant': 11, 'instance': 6,
                                          'witness': 5, 'mulc': 5, 'addc': 2}
    assert run.n_asserts == 2
    assert run.n_live_wires() == 0 and run.queue_len(0) == 0 and run.queue_len(1) == 0


def test_example_op_histograms():
    # acceptance numbers for the current examples.rs / boolean_examples.rs relations
    run = OracleRun(buffers=golden_buffers('arith_101_correct'))
    assert Counter(run.trace_kinds()) == {'copy': 147, 'mul': 57, 'add': 43, 'constant': 11, 'instance': 6,
                                          'witness': 6, 'mulc': 5, 'addc': 2}
    assert run.n_asserts == 6
    run = OracleRun(buffers=golden_buffers('bool_correct'))
    assert Counter(run.trace_kinds()) == {'copy': 49, 'xor': 21, 'and': 19, 'not': 14, 'instance': 10,
                                          'witness': 5, 'constant': 3}
    assert run.n_asserts == 4
    run = OracleRun(buffers=golden_buffers('arith_bn254_correct'))
    assert Counter(run.trace_kinds())['mul'] == 745


def test_exponentiation_known_answers():
    # rust/src/consumers/evaluator.rs:950-984 test_exponentiation
    assert oracle_exp(2, 2206000150907221872269901214599500635,
                      16249742125730185677094195492597105093) == 5834907326474057072663503101785122138
    assert oracle_exp(42, 100, 101) == 1
    # and against Python's pow on a few more
    for b, e, m in [(3, 65537, 2 ** 61 - 1), (123456789, circuits.BN254_R - 1, circuits.BN254_R), (7, 1, 101)]:
        assert oracle_exp(b, e, m) == pow(b, e, m)


def test_no_gate_and_latch():
    inst, wit, rel = golden_buffers('arith_101_incorrect')
    # no relation at all -> "Did not receive any gate to verify." (evaluator.rs:199-203)
    assert OracleRun(buffers=[inst, wit]).violations == ['Did not receive any gate to verify.']
    # after the first error later messages are ignored (evaluator.rs:213-222)
    run = OracleRun(buffers=[inst, wit, rel, rel])
    assert run.violations == ['Wire_9 (may be weighted) should be 0, while it is not']
    assert len(run.trace_kinds()) == 30


def test_missing_inputs():
    inst, wit, rel = golden_buffers('arith_101_correct')
    assert OracleRun(buffers=[wit, rel]).violations == ['Not enough instance to consume']
    run = OracleRun(buffers=[inst, rel])  # PlaintextBackend panics on a missing witness (evaluator.rs:944-946)
    assert run.panicked and 'Missing witness value' in run.violations[0]


def test_cpu_opt_agrees_with_the_literal_oracle():
    """oracle/cpu_opt.cpp (flat array + 64-bit Montgomery on the recorded tape) against the literal
    restatement: every value and the first failing assert, arithmetic goldens over two fields."""
    import zkinterface_ir_amd as zk
    from helpers import le_values
    from oracle_lib import opt_eval
    for name, p, inst, wit in [('arith_bn254_correct', circuits.BN254_R, [25, 0, 1], [3, 4, 0, 17711]),
                               ('arith_101_correct', 101, [25, 0, 1], [3, 4, 0, 36]),
                               ('arith_101_incorrect', 101, [25, 0, 1], [3, 5, 1, 40])]:
        bufs = golden_buffers(name)
        ev = zk.Evaluator.from_messages(bufs)
        kinds, a, b = ev.tape()
        mod_le = p.to_bytes(32, 'little')
        ff, _, vals = opt_eval(kinds, a, b, ev.constants(), mod_le, le_values(inst, 32), len(inst), le_values(wit, 32),
                               len(wit), 32, 1, 1, dump_lane=0)
        ref = OracleRun(buffers=bufs)
        rv = ref.trace_values()
        assert vals[:len(rv)] == rv
        if ref.violations:
            assert ref.violations == ['Wire_%d (may be weighted) should be 0, while it is not' % int(ev.assert_wires()[int(ff[0])])]
        else:
            assert int(ff[0]) == 0xFFFFFFFF


def test_c2_full_size_digest_fixture_matches_the_oracle():
    """tests/golden/c2_digests.json (made by tests/golden/make_c2_digests.py) is what the oracle computes for the
    BASELINE configs[1] relation: one lane is recomputed here, the GPU tier compares all of them."""
    import hashlib
    import json
    import os
    from helpers import ROOT, oracle_lane
    from zkinterface_ir_amd import workloads
    fx = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'c2_digests.json')))
    wl = workloads.ArithLayered()
    inst, wit = wl.inputs(98)
    lane = 97
    iv = [int.from_bytes(inst[lane, k].tobytes(), 'little') for k in range(wl.n_instance0)]
    wv = [int.from_bytes(wit[lane, k].tobytes(), 'little') for k in range(wl.n_witness)]
    run = oracle_lane(wl.mod_le, iv, wv, wl.relation_messages(with_epilogue=False, free_last=False), wl.width, trace=False)
    vals = [run.get(w) for w in wl.output_wire_ids()]
    assert hashlib.sha256('\n'.join(str(v) for v in vals).encode()).hexdigest() == fx['lanes'][str(lane)]['sha256']
    assert str(vals[0]) == fx['lanes'][str(lane)]['first_output']


def test_all_lanes_fixtures_are_what_their_generator_says():
    """tests/golden/c2_all_lanes.json / c4_all_lanes.json hold one 64-bit hash of the 64 output wires for EVERY lane
    (8192 of C2 = configs[2]'s 8 ranks x 1024; 4096 of C4), made by the fast CPU checkers.  Regenerated here for a
    sample: lanes 0..127 of rank 0 -- among them four of the six lanes the literal oracle's digests pin (0, 1, 96, 97), so
    the chain oracle -> cpu_opt -> all lanes is checked end to end -- and a few lanes of another rank's share."""
    import hashlib
    import json
    import os
    import cpu_checkers
    from helpers import ROOT
    from zkinterface_ir_amd import workloads
    fx = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'c2_all_lanes.json')))
    six = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'c2_digests.json')))
    assert len(fx['hashes']) == 8192 and len(set(fx['hashes'])) == 8192
    wl = workloads.ArithLayered()
    inst, wit = wl.inputs(128)
    out = cpu_checkers.arith_layered_outputs(wl, inst, wit, threads=8)
    cpu_checkers.check_against_all_lanes_fixture(fx, out)
    for lane in (0, 1, 96, 97):
        vals = [int.from_bytes(out[lane, t].tobytes(), 'little') for t in range(wl.n_out)]
        assert hashlib.sha256('\n'.join(str(v) for v in vals).encode()).hexdigest() == six['lanes'][str(lane)]['sha256'], lane
    inst, wit = wl.inputs(8, 3 * 1024)           # rank 3 of configs[2]
    cpu_checkers.check_against_all_lanes_fixture(fx, cpu_checkers.arith_layered_outputs(wl, inst, wit, threads=8), 3 * 1024)
    # a damaged output does not pass
    bad = out.copy()
    bad[5, 0, 0] ^= 1
    import pytest
    with pytest.raises(AssertionError, match='lane 5'):
        cpu_checkers.check_against_all_lanes_fixture(fx, bad)
