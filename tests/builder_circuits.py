"""The circuits the reference's producer tests build (rust/src/producers/builder.rs:727-1175,
rust/src/producers/from_r1cs.rs:176-217), restated against zkinterface_ir_amd.builder.  Values, call order and the
calls that must fail are the reference's; every circuit evaluates with zero violations there."""
import pytest

from zkinterface_ir_amd.builder import (ARITH, FOR_FUNCTION_SWITCH, BuilderError, GateBuilder, Header, MemorySink)
from zkinterface_ir_amd.from_r1cs import FromR1CSConverter

EXAMPLE_HEADER = lambda: Header((101).to_bytes(4, 'little'))  # producers/examples.rs:11-13,31-36


def v(x):
    return bytes([x])


def new_builder(sink=None):
    return GateBuilder(sink or MemorySink(), EXAMPLE_HEADER(), ARITH, FOR_FUNCTION_SWITCH)


def with_function(sink=None):  # builder.rs:727-804
    b = new_builder(sink)
    fb = b.new_function_builder('custom_sub', 2, 4)
    i = fb.input_wire_ids()
    n2 = fb.create_gate(('mulc', i[2], v(100)))
    n3 = fb.create_gate(('mulc', i[3], v(100)))
    o0 = fb.create_gate(('add', i[0], n2))
    o1 = fb.create_gate(('add', i[1], n3))
    b.push_function(fb.finish([o0, o1]))
    with pytest.raises(BuilderError, match='already exists'):
        b.push_function(('custom_sub', 0, 0, 0, 0, []))
    ids = [b.create_gate(('constant', v(x))) for x in (40, 30, 10, 5)]
    out = b.create_complex_gate(('call', 'custom_sub', ids))
    from zkinterface_ir_amd.builder import expand_wirelist
    out = expand_wirelist(out)
    assert len(out) == 2
    w0 = b.create_gate(('witness', v(30)))
    w1 = b.create_gate(('witness', v(25)))
    nw0 = b.create_gate(('mulc', w0, v(100)))
    nw1 = b.create_gate(('mulc', w1, v(100)))
    r0 = b.create_gate(('add', out[0], nw0))
    r1 = b.create_gate(('add', out[1], nw1))
    b.create_gate(('assert_zero', r0))
    b.create_gate(('assert_zero', r1))
    with pytest.raises(BuilderError, match='does not exist'):
        b.create_complex_gate(('call', 'unknown_function', [ids[0]]))
    return b.finish()


def with_several_functions(sink=None):  # builder.rs:806-893
    from zkinterface_ir_amd.builder import expand_wirelist
    b = new_builder(sink)
    fb = b.new_function_builder('witness_square', 1, 0)
    w = fb.create_gate(('witness', None))
    o = fb.create_gate(('mul', w, w))
    b.push_function(fb.finish([o]))

    fb = b.new_function_builder('sub_instance_witness_square', 1, 0)
    inst = fb.create_gate(('instance', None))
    with pytest.raises(BuilderError, match='has 0 inputs and is called with 1 inputs'):
        fb.create_complex_gate(('call', 'witness_square', [inst]))
    with pytest.raises(BuilderError, match='does not exist'):
        fb.create_complex_gate(('call', 'test', [inst]))
    sq = expand_wirelist(fb.create_complex_gate(('call', 'witness_square', [])))
    neg = fb.create_gate(('mulc', sq[0], v(100)))
    o = fb.create_gate(('add', inst, neg))
    b.push_function(fb.finish([o]))

    with pytest.raises(BuilderError, match='has 1 instances and is called with 0 instances'):
        b.create_complex_gate(('call', 'sub_instance_witness_square', []), [], [v(5)])
    with pytest.raises(BuilderError, match='has 1 witnesses and is called with 0 witnesses'):
        b.create_complex_gate(('call', 'sub_instance_witness_square', []), [v(25)], [])
    out = expand_wirelist(b.create_complex_gate(('call', 'sub_instance_witness_square', []), [v(25)], [v(5)]))
    assert len(out) == 1
    b.create_gate(('assert_zero', out[0]))
    return b.finish()


def _sub_and_add(b, add_extra_witness):
    fb = b.new_function_builder('custom_sub', 2, 2)
    i = fb.input_wire_ids()
    inst = fb.create_gate(('instance', None))
    wit = fb.create_gate(('witness', None))
    ni = fb.create_gate(('mulc', inst, v(100)))
    nw = fb.create_gate(('mulc', wit, v(100)))
    o0 = fb.create_gate(('add', i[0], ni))
    o1 = fb.create_gate(('add', i[1], nw))
    b.push_function(fb.finish([o0, o1]))

    fb = b.new_function_builder('custom_add', 2, 2)
    i = fb.input_wire_ids()
    inst = fb.create_gate(('instance', None))
    wit = fb.create_gate(('witness', None))
    o0 = fb.create_gate(('add', i[0], inst))
    o1 = fb.create_gate(('add', i[1], wit))
    if add_extra_witness:
        w2 = fb.create_gate(('witness', None))
        fb.create_gate(('assert_zero', w2))
    b.push_function(fb.finish([o0, o1]))


def _assert_equal_witness(b):
    fb = b.new_function_builder('assert_equal_witness', 0, 1)
    i = fb.input_wire_ids()
    wit = fb.create_gate(('witness', None))
    nw = fb.create_gate(('mulc', wit, v(100)))
    r = fb.create_gate(('add', i[0], nw))
    fb.create_gate(('assert_zero', r))
    b.push_function(fb.finish([]))


def switch_builder(sink=None):  # builder.rs:895-1053
    from zkinterface_ir_amd.builder import expand_wirelist
    b = new_builder(sink)
    _sub_and_add(b, add_extra_witness=True)
    _assert_equal_witness(b)
    in0 = b.create_gate(('constant', v(10)))
    in1 = b.create_gate(('constant', v(15)))
    cond = b.create_gate(('constant', v(1)))
    sb = b.new_switch_builder(2)
    with pytest.raises(BuilderError, match='does not exist'):
        sb.create_branch_from('unknown_function', [in0, in1])
    sb.push_branch(sb.create_branch_from('custom_sub', [in0, in1]), v(0))
    with pytest.raises(BuilderError, match='two cases with the same value'):
        sb.push_branch(sb.create_branch_from('custom_add', [in0, in1]), v(0))
    sb.push_branch(sb.create_branch_from('custom_add', [in0, in1]), v(1))
    switch = sb.finish(cond)
    out = expand_wirelist(b.create_complex_gate(switch, [v(5)], [v(15), v(0)]))
    b.create_complex_gate(('call', 'assert_equal_witness', [out[0]]), [], [v(15)])
    b.create_complex_gate(('call', 'assert_equal_witness', [out[1]]), [], [v(30)])
    with pytest.raises(BuilderError, match='empty switch'):
        b.new_switch_builder(0).finish(cond)
    v55 = b.create_gate(('constant', v(55)))
    cond2 = b.create_gate(('constant', v(60)))
    sb = b.new_switch_builder(0)
    sb.push_branch(sb.create_branch_from('assert_equal_witness', [v55]), v(60))
    b.create_complex_gate(sb.finish(cond2), [], [v(55)])
    sb = b.new_switch_builder(0)
    sb.push_branch(sb.create_branch_from('assert_equal_witness', [v55]), v(60))
    with pytest.raises(BuilderError, match='Switch has 0 witnesses and is called with 0 witnesses'):
        b.create_complex_gate(sb.finish(cond2), [], [])
    return b.finish()


def switch_nested_in_function(sink=None):  # builder.rs:1055-1175
    from zkinterface_ir_amd.builder import expand_wirelist
    b = new_builder(sink)
    _sub_and_add(b, add_extra_witness=False)
    id0 = b.create_gate(('constant', v(40)))
    id1 = b.create_gate(('constant', v(30)))
    cond = b.create_gate(('constant', v(1)))
    fb = b.new_function_builder('function_with_switch', 2, 3)
    i = fb.input_wire_ids()
    sb = b.new_switch_builder(2)
    sb.push_branch(sb.create_branch_from('custom_sub', [i[0], i[1]]), v(0))
    sb.push_branch(sb.create_branch_from('custom_add', [i[0], i[1]]), v(1))
    out = expand_wirelist(fb.create_complex_gate(sb.finish(i[2])))
    b.push_function(fb.finish(out))
    out = expand_wirelist(b.create_complex_gate(('call', 'function_with_switch', [id0, id1, cond]), [v(10)], [v(5)]))
    _assert_equal_witness(b)
    b.create_complex_gate(('call', 'assert_equal_witness', [out[0]]), [], [v(50)])
    b.create_complex_gate(('call', 'assert_equal_witness', [out[1]]), [], [v(35)])
    return b.finish()


BUILDER_CIRCUITS = {'with_function': with_function, 'with_several_functions': with_several_functions,
                    'switch_builder': switch_builder, 'switch_nested_in_function': switch_nested_in_function}


# zkinterface's example (x^2 + y^2 = zz; used by from_r1cs.rs:176-217 with x=3, y=4, zz=25 over GF(101)):
# ids 0 = one, 1..3 = instance x, y, zz, 4..5 = witness xx, yy.
def r1cs_example(sink=None, x=3, y=4, zz=25):
    one = v(1)
    conv = FromR1CSConverter(sink or MemorySink(), 100, [(0, one), (1, v(x)), (2, v(y)), (3, v(zz))], [4, 5])
    conv.ingest_witness([(4, v(x * x % 101)), (5, v(y * y % 101))])
    conv.ingest_constraints([
        ([(1, one)], [(1, one)], [(4, one)]),               # x * x = xx
        ([(2, one)], [(2, one)], [(5, one)]),               # y * y = yy
        ([(0, one)], [(4, one), (5, one)], [(3, one)]),     # 1 * (xx + yy) = zz
    ])
    return conv.finish()


R1CS_EXAMPLE_WIRES = [1, 100, 3, 4, 25, 9, 16]  # from_r1cs.rs:201-214
R1CS_EXAMPLE_STATS = dict(instance_variables=3, witness_variables=2, constants_gates=12, assert_zero_gates=3, add_gates=4,
                          mul_gates=15, instance_messages=1, witness_messages=1, relation_messages=1)  # :247-272
