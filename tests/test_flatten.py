"""The flattened workspace emitted from the tape is equisatisfiable with the original statement and
carries the same values: the reference's own acceptance test for IRFlattener
(rust/src/consumers/flattening.rs:194-252 test_validate_flattening / test_evaluate_flattening)."""
import pytest

import circuits
from helpers import golden_buffers
from oracle_lib import OracleRun
from test_host_tape import INPUTS
import zkinterface_ir_amd as zk
from zkinterface_ir_amd import flatten as fl
from zkinterface_ir_amd import sieve_writer as sw


@pytest.mark.parametrize('name', sorted(INPUTS))
def test_flattened_relation_re_evaluates_like_the_original(name):
    p, inst, wit = INPUTS[name]
    bufs = golden_buffers(name)
    ev = zk.Evaluator.from_messages(bufs)
    mod_le = sw.int_to_le(p)
    rel, positions = fl.flatten(ev, mod_le, boolean=(p == 2))
    fi, fw = fl.flattened_inputs(mod_le, [sw.int_to_le(v) for v in inst], [sw.int_to_le(v) for v in wit], positions)
    orig = OracleRun(buffers=bufs)
    flat = OracleRun(buffers=[fi, fw, rel])
    assert (flat.violations == []) == (orig.violations == [])
    # simple gates only: one backend call per gate, plus the `copy` the reference inserts in front of an
    # unweighted AssertZero (evaluator.rs:352-356) -- drop those and the two traces coincide
    k1, _, _ = ev.tape()
    fv, fk = flat.trace_values(), flat.trace_kinds()
    vals, kinds_seen, pos = [], [], 0
    for k in k1:
        if pos >= len(fv):
            break
        if int(k) == 9:
            assert fk[pos] == 'copy'
            pos += 1
        else:
            vals.append(fv[pos])
            kinds_seen.append(fk[pos])
            pos += 1
    n = len(orig.trace_values())
    assert vals[:n] == orig.trace_values()
    assert kinds_seen[:n] == orig.trace_kinds()
    if not orig.violations:
        assert pos == len(fv) and len(vals) == n
    # the product records the flattened relation too, with the same asserts
    ev2 = zk.Evaluator.from_messages([fi, fw, rel])
    assert ev2.host_violations() == []
    assert ev2.n_asserts == ev.n_asserts and ev2.n_value_ops == ev.n_value_ops + ev.n_asserts


def test_flatten_tool_writes_an_equisatisfiable_workspace(tmp_path):
    """`flatten <workspace> --out <dir>` (cli.rs:442-472) through the file sink."""
    import io
    from zkinterface_ir_amd import cli
    for name, sat in (('arith_101_correct', True), ('arith_101_incorrect', False), ('bool_correct', True)):
        src, dst = tmp_path / (name + '_src'), tmp_path / (name + '_flat')
        src.mkdir()
        for k, b in enumerate(golden_buffers(name)):
            (src / ('%03d.sieve' % k)).write_bytes(b)
        err = io.StringIO()
        assert cli.main(['flatten', str(src), '--out', str(dst)], err=err) == 0, err.getvalue()
        assert sorted(p.name for p in dst.iterdir()) == ['000_instance.sieve', '001_witness.sieve', '002_relation.sieve']
        flat = OracleRun(files=[str(dst / n) for n in ('000_instance.sieve', '001_witness.sieve', '002_relation.sieve')])
        assert (flat.violations == []) == sat
        # only simple gates, and the validator accepts the result
        ev = zk.Evaluator()
        ev.set_option('validate', 'prover')
        ev.set_option('metrics', '1')
        ev.ingest_paths([str(dst)])
        assert ev.validator_violations() == []
        gs = ev.stats()['gate_stats']
        assert gs['functions_called'] == gs['switches'] == gs['for_loops'] == 0
    assert cli.main(['flatten', str(src), '--out', str(tmp_path / 'x.sieve')], err=io.StringIO()) == 1


@pytest.mark.parametrize('name,gate_set,absent', [
    ('arith_101_correct', '@add,@mul', ('add_constant_gates', 'mul_constant_gates')),
    ('arith_101_incorrect', '@add,@mul,@mulc', ('add_constant_gates',)),
    ('bool_correct', '@xor,@and,@add', ('not_gates', 'add_constant_gates')),
    ('bool_correct', '@add,@mul,@addc', ('xor_gates', 'and_gates', 'not_gates')),
])
def test_expand_definable_rewrites_the_missing_gates(tmp_path, name, gate_set, absent):
    """`expand-definable --gate-set ...` (cli.rs:515-555, consumers/exp_definable.rs): the flattened relation uses
    only the target gateset and is equisatisfiable with the original statement."""
    import io
    from zkinterface_ir_amd import cli
    src, dst = tmp_path / 'src', tmp_path / 'out'
    src.mkdir()
    bufs = golden_buffers(name)
    for k, b in enumerate(bufs):
        (src / ('%03d.sieve' % k)).write_bytes(b)
    err = io.StringIO()
    assert cli.main(['expand-definable', str(src), '--gate-set', gate_set, '--out', str(dst)], err=err) == 0, err.getvalue()
    ev = zk.Evaluator()
    ev.set_option('metrics', '1')
    ev.set_option('validate', 'prover')
    ev.ingest_paths([str(dst)])
    gs = ev.stats()['gate_stats']
    assert all(gs[k] == 0 for k in absent), gs
    flat = OracleRun(files=[str(dst / n) for n in ('000_instance.sieve', '001_witness.sieve', '002_relation.sieve')])
    assert (flat.violations == []) == (OracleRun(buffers=bufs).violations == [])


def test_expand_definable_refuses_what_the_reference_panics_on(tmp_path):
    from zkinterface_ir_amd import flatten as fl2
    ev = zk.Evaluator.from_messages(golden_buffers('bool_correct'))
    with pytest.raises(ValueError, match='Cannot replace XOR by ADD if ADD is not supported.'):
        fl2.flatten(ev, bytes([2]), boolean=True, gate_mask=fl2.parse_gate_set('@and,@not'))
    with pytest.raises(ValueError, match='Unable to parse the following gateset'):
        fl2.parse_gate_set('@add,@nope')
    assert fl2.parse_gate_set('arithmetic') == 0xF and fl2.parse_gate_set(' @xor, @and,') == 0x300
