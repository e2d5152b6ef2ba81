"""Producer side (builder.py, from_r1cs.py) against the reference's own producer tests: the four GateBuilder
circuits of rust/src/producers/builder.rs:727-1175 evaluate with zero violations, replace_output_wires reproduces
rust/src/structs/gates.rs:856-990, the sinks behave like rust/src/producers/sink.rs:147-223, and the R1CS example
gives the wire values and gate counts of rust/src/producers/from_r1cs.rs:176-286.  Evaluation here is the oracle
plus the product's host recording; tests/test_gpu_parity.py replays the same circuits on the GPU."""
import os

import pytest

import zkinterface_ir_amd as zk
from builder_circuits import (BUILDER_CIRCUITS, R1CS_EXAMPLE_STATS, R1CS_EXAMPLE_WIRES, r1cs_example, v, new_builder)
from oracle_lib import OracleRun
from validator_ref import GATE_STAT_FIELDS
from zkinterface_ir_amd import builder as bld


@pytest.mark.parametrize('name', sorted(BUILDER_CIRCUITS))
def test_builder_circuit_is_true_and_compliant(name):
    sink = BUILDER_CIRCUITS[name]()
    ref = OracleRun(buffers=sink.buffers())
    assert ref.violations == []
    ev = zk.Evaluator()
    ev.set_option('validate', 'prover')
    for b in sink.buffers():
        ev.ingest_message(b)
    assert ev.host_violations() == []
    assert ev.validator_violations() == []
    assert [zk.KIND_NAMES[k] for k in ev.tape()[0] if k != 9] == ref.trace_kinds()


def test_r1cs_example_wire_values_and_stats():  # from_r1cs.rs:176-286
    sink = r1cs_example()
    ref = OracleRun(buffers=sink.buffers())
    assert ref.violations == []
    assert [ref.get(w) for w in range(7)] == R1CS_EXAMPLE_WIRES
    ev = zk.Evaluator()
    ev.set_option('metrics', '1')
    ev.set_option('validate', 'prover')
    for b in sink.buffers():
        ev.ingest_message(b)
    want = dict.fromkeys(GATE_STAT_FIELDS, 0)
    want.update(R1CS_EXAMPLE_STATS)
    assert ev.stats() == {'field_characteristic': [101], 'field_degree': 1, 'gate_stats': want, 'functions': {}}
    assert ev.validator_violations() == []
    # a wrong witness makes exactly the first constraint fail
    bad = OracleRun(buffers=r1cs_example(x=3, y=4, zz=26).buffers())
    assert bad.violations == ['Wire_%d (may be weighted) should be 0, while it is not' % 35]


def test_r1cs_converter_errors():
    from zkinterface_ir_amd.from_r1cs import FromR1CSConverter
    conv = FromR1CSConverter(bld.MemorySink(), 100, [(0, v(1)), (1, v(3))], [2])
    with pytest.raises(bld.BuilderError, match='The WireId 9 has not been defined yet.'):
        conv.ingest_constraints([([(9, v(1))], [], [])])
    with pytest.raises(bld.BuilderError, match='The ZKI witness id 7 does not exist.'):
        conv.ingest_witness([(7, v(1))])
    with pytest.raises(bld.BuilderError, match='field_maximum must be provided'):
        FromR1CSConverter(bld.MemorySink(), None, [], [])


def test_replace_output_wires():  # structs/gates.rs:856-915
    gates = [('instance', 4), ('witness', 5), ('constant', 6, v(15)), ('add', 7, 4, 5), ('free', 4, 5), ('mul', 8, 6, 7),
             ('call', 'custom', [(9, 12)], [(6, 8)]), ('assert_zero', 12),
             ('switch', 6, [13, 14, 15], [v(2), v(5)], [('call', 'function_branch0', [(6, 8)]),
                                                        ('call', 'function_branch1', [10])])]
    bld.replace_output_wires(gates, [6, 11, 12, 15])
    assert gates == [('instance', 4), ('witness', 5), ('constant', 0, v(15)), ('add', 7, 4, 5), ('free', 4, 5),
                     ('mul', 8, 0, 7), ('call', 'custom', [9, 10, 1, 2], [0, 7, 8]), ('assert_zero', 2),
                     ('switch', 0, [13, 14, 3], [v(2), v(5)], [('call', 'function_branch0', [0, 7, 8]),
                                                               ('call', 'function_branch1', [10])])]


def test_replace_output_wires_with_for():  # structs/gates.rs:917-963
    loop = ('for', 'i', 10, 12, [(10, 12)], ('anon', [('name', 'i')], [], 0, 1, [('witness', 0)]))
    gates = [loop, ('xor', 13, 10, 11), ('assert_zero', 13)]
    bld.replace_output_wires(gates, [10, 11, 12, 13])
    assert gates == [loop, ('xor', 13, 10, 11), ('assert_zero', 13), ('copy', 0, 10), ('copy', 1, 11), ('copy', 2, 12),
                     ('copy', 3, 13)]


def test_replace_output_wires_with_forbidden_free():  # structs/gates.rs:965-990
    for free in (('free', 7, 9), ('free', 4, None)):
        gates = [('xor', 2, 4, 6), ('and', 7, 4, 6), free, ('xor', 8, 3, 5), ('xor', 9, 7, 8), ('and', 10, 3, 5),
                 ('not', 11, 10)]
        with pytest.raises(bld.BuilderError, match='forbidden to free an output wire'):
            bld.replace_output_wires(gates, [8, 4])


def test_replace_wire_in_wirelist():  # structs/wire.rs:252-267
    wl = [(0, 2), 5]
    assert bld._replace_in_wirelist(wl, 4, 14) == [(0, 2), 5]
    assert bld._replace_in_wirelist(wl, 5, 15) == [0, 1, 2, 15]
    assert bld._replace_in_wirelist(wl, 1, 14) == [0, 14, 2, 5]
    assert bld.wirelist_len([(0, 2), 5]) == 4


def test_files_sink(tmp_path):  # producers/sink.rs:147-223
    from circuits import arith_example
    ws = str(tmp_path / 'test_sink')
    sink = bld.FilesSink.new_clean(ws)
    names = ['000_instance.sieve', '001_witness.sieve', '002_relation.sieve']
    sizes = lambda: [os.path.getsize(os.path.join(ws, n)) for n in sorted(os.listdir(ws))]
    assert sorted(os.listdir(ws)) == names and sizes() == [0, 0, 0]
    inst, wit, rel = arith_example()
    last = sizes()
    for _ in range(2):
        sink.push_instance_message(inst)
        sink.push_witness_message(wit)
        sink.push_relation_message(rel)
        now = sizes()
        assert sorted(os.listdir(ws)) == names and all(a < b for a, b in zip(last, now))
        last = now
    sink.close()
    ev = zk.Evaluator()
    ev.set_option('metrics', '1')
    ev.ingest_paths(sink.paths())
    gs = ev.stats()['gate_stats']
    assert (gs['instance_messages'], gs['witness_messages'], gs['relation_messages']) == (2, 2, 2)
    (tmp_path / 'test_sink' / 'notes.txt').write_text('kept')
    bld.clean_workspace(ws)
    assert os.listdir(ws) == ['notes.txt']


def test_message_builder_flushes_at_max_len():  # builder.rs:77-103,118-133
    b = new_builder()
    b.msg_build.max_len = 10
    prev = b.create_gate(('instance', v(1)))
    for k in range(34):
        prev = b.create_gate(('addc', prev, v(1)))
        b.push_witness_value(v(k))
    fb = b.new_function_builder('f', 1, 1)
    o = fb.create_gate(('copy', fb.input_wire_ids()[0]))
    b.push_function(fb.finish([o]))
    out = b.create_complex_gate(('call', 'f', [prev]))
    b.create_gate(('assert_zero', b.create_gate(('addc', out[0], v(101 - 35)))))
    sink = b.finish()
    ev = zk.Evaluator()
    ev.set_option('metrics', '1')
    ev.set_option('validate', 'prover')
    for buf in sink.buffers():
        ev.ingest_message(buf)
    gs = ev.stats()['gate_stats']
    # 35 gates -> 3 full relation messages of 10, then f (1 body gate) + call + addc + assert in the last one
    assert (gs['relation_messages'], gs['witness_messages'], gs['instance_messages']) == (4, 4, 1)
    assert gs['add_constant_gates'] == 35 and gs['functions_called'] == 1
    # 34 witness values nobody consumes: the validator says so, the evaluator does not mind
    assert ev.validator_violations() == ['Too many Witness values (34 not consumed)']
    assert OracleRun(buffers=sink.buffers()).violations == []


def test_gateset_and_feature_strings():  # structs/relation.rs:181-225,255-281
    assert bld.create_gateset_string(bld.ARITH) == 'arithmetic'
    assert bld.create_gateset_string(bld.BOOL) == 'boolean'
    assert bld.create_gateset_string(bld.ADD | bld.MULC | bld.XOR) == '@add,@mulc,@xor,'
    assert bld.create_gateset_string(bld.NOT | bld.AND) == '@not,@and,'
    assert bld.create_feature_string(bld.SIMPLE) == 'simple'
    assert bld.create_feature_string(bld.FOR_FUNCTION_SWITCH) == '@for,@switch,@function,'
    assert bld.create_feature_string(bld.FUNCTION) == '@function,'
