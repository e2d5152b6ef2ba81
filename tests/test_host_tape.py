"""CPU tier: product host logic (reader, Evaluator, recording backend, scheduler, C ABI
surface) checked against the oracle.  No GPU call is made here: the scheduled device
program is interpreted by tests/program_sim.py."""
import ctypes
import os
import re

import numpy as np
import pytest

import circuits
import program_sim
from helpers import REF_EXAMPLES, ROOT, golden_buffers, oracle_lane
from oracle_lib import OracleRun
import zkinterface_ir_amd as zk
from zkinterface_ir_amd import workloads

INPUTS = {
    'ref_examples': (101, [25, 0, 1], [3, 4, 36]),
    'arith_101_correct': (101, [25, 0, 1], [3, 4, 0, 36]),
    'arith_101_incorrect': (101, [25, 0, 1], [3, 5, 1, 40]),
    'bool_correct': (2, [0, 0, 0, 0, 0, 1, 0, 1], [1, 0, 1, 0, 0]),
    'bool_incorrect': (2, [0, 0, 0, 0, 0, 1, 0, 1], [1, 1, 1, 0, 0]),
    'arith_bn254_correct': (circuits.BN254_R, [25, 0, 1], [3, 4, 0, 17711]),
}


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'zkgpu.h')).read()
    declared = set(re.findall(r'\b(zkgpu_[a-z_0-9]+)\s*\(', header))
    declared.discard('zkgpu_session')
    L = ctypes.CDLL(zk.LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert declared == set(zk.exported_symbols()), declared ^ set(zk.exported_symbols())
    assert b'gfx950' in zk.lib().zkgpu_version()


@pytest.mark.parametrize('name', sorted(INPUTS))
def test_tape_matches_oracle_trace(name):
    bufs = golden_buffers(name)
    ev = zk.Evaluator.from_messages(bufs)
    kinds, a, b = ev.tape()
    mine = [zk.KIND_NAMES[k] for k in kinds if k != 9]
    ref = OracleRun(buffers=bufs)
    ref_kinds = ref.trace_kinds()
    # the reference stops at the first failing assert; the tape always records the whole relation
    assert mine[:len(ref_kinds)] == ref_kinds
    if not ref.violations:
        assert len(mine) == len(ref_kinds) and ev.n_asserts == ref.n_asserts
    assert ev.host_violations() == []


@pytest.mark.parametrize('retain', [True, False])
@pytest.mark.parametrize('name', sorted(INPUTS))
def test_scheduled_program_matches_oracle(name, retain):
    p, inst, wit = INPUTS[name]
    bufs = golden_buffers(name)
    ev = zk.Evaluator.from_messages(bufs)
    ev.finalize(retain_all=retain)
    ops, launches, consts, slot_of = ev.schedule_dump()
    info = ev.schedule_info()
    ref = OracleRun(buffers=bufs)
    for shuffle in (None, 7):
        slots, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p,
                                                   inst, wit, shuffle_seed=shuffle)
        assert not noncanon
        assert (ff is None) == (ref.violations == [])
        if retain:
            kinds, _, _ = ev.tape()
            vals = [program_sim.from_device_form(slots[slot_of[i]], p, info['words_per_const'])
                    for i in range(len(kinds)) if kinds[i] != 9]
            rv = ref.trace_values()
            assert vals[:len(rv)] == rv
    if not retain:
        assert info['slots'] < ev.n_value_ops  # liveness actually reuses slots


@pytest.mark.parametrize('p', [circuits.P320, circuits.BLS12_381_Q, circuits.P448, circuits.P512])
def test_fields_wider_than_256_bits(p):
    """five- to eight-limb fields (up to 512 bits) through recording, scheduling and the interpreter"""
    inst, wit, rel = circuits.arith_example(p)
    bufs = [inst, wit, rel]
    ref = OracleRun(buffers=bufs, width=64)
    assert ref.violations == []
    for retain in (True, False):
        ev = zk.Evaluator.from_messages(bufs)
        assert ev.host_violations() == [] and ev.elem_bytes == 8 * ((p.bit_length() + 63) // 64)
        ev.finalize(retain_all=retain)
        slots, ff, slot_of, info = _sim_lane(ev, p, [25, 0, 1], [3, 4, 0, 17711])
        assert ff is None and info['words_per_const'] == ev.elem_bytes // 4
        if retain:
            kinds, _, _ = ev.tape()
            vals = [program_sim.from_device_form(slots[slot_of[i]], p, info['words_per_const'])
                    for i in range(len(kinds)) if kinds[i] != 9]
            assert vals == ref.trace_values()
    # (beyond 512 bits: the any-modulus kernels, tests/test_any_modulus.py)
    too_wide = circuits.arith_example(2 ** 4100 + 1)[2]
    assert any('wider than 4096 bits' in m for m in zk.Evaluator.from_messages([too_wide]).host_violations())


def test_reader_file_ordering_and_framing(tmp_path):
    # Source::from_filenames ordering (source.rs:69-89): name sort, then instance < witness < relation
    ev = zk.Evaluator()
    ev.ingest_paths(list(reversed(REF_EXAMPLES)))
    assert ev.host_violations() == []
    assert ev.n_value_ops == 208 and ev.n_asserts == 2
    # a directory works too, and several messages may share one buffer / file
    d = tmp_path / 'ws'
    d.mkdir()
    inst, wit, rel = golden_buffers('arith_101_correct')
    (d / '000_instance.sieve').write_bytes(inst)
    (d / '001_witness.sieve').write_bytes(wit)
    (d / '002_relation.sieve').write_bytes(rel)
    (d / 'notes.txt').write_bytes(b'ignored')
    ev2 = zk.Evaluator()
    ev2.ingest_paths([str(d)])
    assert ev2.n_value_ops == 277
    ev3 = zk.Evaluator.from_messages([inst + wit + rel + b'\x00\x00\x00\x00' + rel])  # size 0 = end marker
    assert ev3.n_value_ops == 277


def test_recording_errors_use_reference_strings():
    inst, wit, rel = golden_buffers('arith_101_correct')
    assert zk.Evaluator.from_messages([inst, wit]).host_violations() == ['Did not receive any gate to verify.']
    assert zk.Evaluator.from_messages([wit, rel]).host_violations() == ['Not enough instance to consume']
    v = zk.Evaluator.from_messages([inst, rel]).host_violations()
    assert len(v) == 1 and 'Missing witness value' in v[0]
    # same strings as the oracle for structurally broken relations
    from zkinterface_ir_amd import sieve_writer as sw
    mod = bytes([101])
    cases = {
        'ssa': [('constant', 0, b'\x01'), ('constant', 0, b'\x02')],
        'missing': [('add', 2, 0, 1)],
        'unknown_fn': [('call', 'nope', [0], [])],
        'bad_range': [('constant', 5, b'\x01'), ('anoncall', [(3, 3)], [], 0, 0, [('constant', 0, b'\x01')])],
        'free_twice': [('constant', 0, b'\x01'), ('free', 0, None), ('free', 0, None)],
    }
    for name, gates in cases.items():
        rel = sw.write_relation(mod, 'arithmetic', '@function,@for,@switch', [], gates)
        ref = OracleRun(buffers=[rel]).violations
        got = zk.Evaluator.from_messages([rel]).host_violations()
        assert got == ref and len(ref) == 1, (name, got, ref)
    fn = [('f', 1, 2, 0, 0, [('add', 0, 1, 2)])]
    rel = sw.write_relation(mod, 'arithmetic', '@function', fn,
                            [('constant', 1, b'\x01'), ('call', 'f', [0], [1])])
    ref = OracleRun(buffers=[rel]).violations
    assert zk.Evaluator.from_messages([rel]).host_violations() == ref
    assert 'Wrong number of input variables' in ref[0]


def test_synthetic_workload_small_against_oracle():
    wl = workloads.ArithLayered(W=32, D=6, n_instance0=4, n_out=3)
    probe = wl.relation_messages(with_epilogue=False, free_last=False)
    inst, wit = wl.inputs(3)
    width = wl.width
    outs = np.zeros((3, wl.n_out, width), dtype=np.uint8)
    for lane in range(3):
        iv = [int.from_bytes(inst[lane, k].tobytes(), 'little') for k in range(wl.n_instance0)]
        wv = [int.from_bytes(wit[lane, k].tobytes(), 'little') for k in range(wl.n_witness)]
        assert all(v < wl.p for v in iv + wv)
        run = oracle_lane(wl.mod_le, iv, wv, probe, width, trace=False)
        assert run.violations == []
        for t, wid in enumerate(wl.output_wire_ids()):
            outs[lane, t] = np.frombuffer(run.get(wid).to_bytes(width, 'little'), dtype=np.uint8)
    bad = wl.set_expected_outputs(inst, outs, corrupt_every=2)
    assert bad == 2
    msgs = wl.relation_messages()
    for lane in range(3):
        iv = [int.from_bytes(inst[lane, k].tobytes(), 'little') for k in range(wl.n_instance)]
        wv = [int.from_bytes(wit[lane, k].tobytes(), 'little') for k in range(wl.n_witness)]
        run = oracle_lane(wl.mod_le, iv, wv, msgs, width, trace=False)
        assert (run.violations == []) == (lane % 2 != 0), (lane, run.violations)
        if not run.violations:
            assert run.n_live_wires() == 0  # the relation frees everything it creates
    # the product records exactly the relation's gates (no structured gates here)
    ev = zk.Evaluator()
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in msgs:
        ev.ingest_message(m)
    assert ev.host_violations() == []
    assert ev.n_value_ops == 32 + 32 * 6 + 3 * 4 and ev.n_asserts == 3
    ev.finalize()
    info = ev.schedule_info()
    assert info['slots'] <= 2 * 32 + 16


def test_switch_ladder_becomes_one_entry_only_over_a_prime_field():
    """A Switch weight is 1 - (case - cond)^(p-1) (evaluator.rs:823-839).  The production schedule replaces the
    352-product ladder of a 254-bit prime field by one `x != 0` entry; the recording, the retain_all schedule and a
    composite characteristic keep the ladder.  Values are checked by the simulator tests above and on the GPU."""
    sizes = {}
    for p in (circuits.BN254_R, 7 * 13):
        _inst, _wit, rel = circuits.arith_example(p)
        for name, kw, retain in (('on', {}, False), ('off', {'fermat': 0}, False), ('retain', {}, True)):
            ev = zk.Evaluator()
            for k, v in kw.items():
                ev.set_option(k, str(v))
            ev.declare_inputs(3, 4)
            ev.ingest_message(rel)
            assert ev.host_violations() == []
            n_tape = ev.n_value_ops
            ev.finalize(retain_all=retain)
            # (entries that do something: a strand pads with no-ops to keep a chain on one wave, csrc/schedule.cpp)
            sizes[(p, name)] = (int(((ev.schedule_dump()[0][:, 1] & 0xFF) != 0).sum()), ev.schedule_info()['levels'], n_tape)
    big = circuits.BN254_R
    assert sizes[(big, 'on')][2] == sizes[(big, 'off')][2] == 965          # the recording is the reference's trace
    assert sizes[(big, 'on')][0] < 120 and sizes[(big, 'on')][1] < 60
    assert sizes[(big, 'off')][0] > 600 and sizes[(big, 'off')][1] > 390
    assert sizes[(big, 'retain')][0] >= 965
    assert sizes[(91, 'on')] == sizes[(91, 'off')]                            # 91 = 7 * 13: no shortcut


def test_trait_level_ladder_hint():
    """A caller that drives the ZKBackend entry points itself (the Rust Evaluator of INTEGRATION.md) can name the
    exponent ladder of a Switch weight with zkgpu_backend_ladder; the schedule then shrinks exactly as with the
    bundled Evaluator, and both forms give weight 1 for cond == case and 0 otherwise."""
    p = circuits.BN254_R
    minus_one = (p - 1).to_bytes(32, 'little')

    def record(hint):
        ev = zk.Evaluator()
        ev.backend_set_field(p.to_bytes(32, 'little'))
        cond = ev.backend_witness(0)
        case = ev.backend_constant(bytes([7]))
        base = ev.backend_add(case, ev.backend_mul_constant(cond, minus_one))
        first = ev.tape_len
        bits = bin(p - 1)[2:]                      # square-and-multiply, most significant bit first
        acc = ev.backend_copy(base)
        for b in bits[1:]:
            acc = ev.backend_multiply(acc, acc)
            if b == '1':
                acc = ev.backend_multiply(acc, base)
        if hint:
            ev.backend_ladder(first, base, acc)
        weight = ev.backend_add_constant(ev.backend_mul_constant(acc, minus_one), bytes([1]))
        # assert weight * (cond - 7) == 0 and (1 - weight) * 1 == 0 only when cond == 7: check the weight directly
        ev.backend_assert_zero(ev.backend_add_constant(weight, minus_one), 0)      # fails unless weight == 1
        ev.finalize()
        return ev
    plain, hinted = record(False), record(True)
    from helpers import working_entries
    assert working_entries(plain) > 200 and working_entries(hinted) < 12
    for ev in (plain, hinted):
        ops, launches, consts, _ = ev.schedule_dump()
        info = ev.schedule_info()
        for cond, ok in ((7, True), (8, False), (0, False), (p - 1, False)):
            _, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p, [], [cond])
            assert not noncanon and (ff is None) == ok, (cond, ff)
    with pytest.raises(zk.ZkGpuError, match='not a range of recorded calls'):
        hinted.backend_ladder(10 ** 9, 0, 1)


def test_bogus_and_overlapping_ladder_hints_leave_the_schedule_alone():
    """zkgpu_backend_ladder is a public entry: a range that is not the reference's square-and-multiply recursion over
    p - 1 (evaluator.rs:801-820), two hints over the same calls, or a hint under is_boolean (where `as_mul` is `and`)
    must not change the program -- the schedule is the one of the unhinted recording."""
    p = circuits.BN254_R
    minus_one = (p - 1).to_bytes(32, 'little')

    def record(mode):
        ev = zk.Evaluator()
        ev.backend_set_field(p.to_bytes(32, 'little'), 1, mode == 'boolean')
        cond = ev.backend_witness(0)
        base = ev.backend_add(ev.backend_constant(bytes([7])), ev.backend_mul_constant(cond, minus_one))
        first = ev.tape_len
        bits = bin(p - 1)[2:]
        acc = ev.backend_copy(base)
        for k, b in enumerate(bits[1:]):
            acc = ev.backend_multiply(acc, acc)
            if b == '1' or (mode == 'extra_multiply' and k == 5):     # one multiply too many: not base^(p-1)
                acc = ev.backend_multiply(acc, base)
        if mode == 'wrong_base':
            ev.backend_ladder(first, cond, acc)
        elif mode == 'short_range':
            ev.backend_ladder(first + 2, base, acc)
        elif mode == 'overlap':
            ev.backend_ladder(first, base, acc)
            ev.backend_ladder(first, base, acc)
        elif mode != 'none':
            ev.backend_ladder(first, base, acc)
        weight = ev.backend_add_constant(ev.backend_mul_constant(acc, minus_one), bytes([1]))
        ev.backend_assert_zero(ev.backend_add_constant(weight, minus_one), 0)
        ev.finalize()
        from helpers import working_entries
        return working_entries(ev)
    plain = record('none')
    assert record('good') < 12 < plain
    for mode in ('wrong_base', 'short_range', 'overlap', 'boolean'):
        assert record(mode) == plain, mode
    assert record('extra_multiply') >= plain


def test_layered_program_with_pair_entries_against_oracle():
    """The production schedule of a layered relation (gate fusion, pair entries for producers with two readers in
    one level, shared-operand order) interpreted entry by entry: the verdict and every surviving output wire equal
    the oracle's, with and without each transformation; every index the kernels would use is validated."""
    wl = workloads.ArithLayered(W=48, D=7, n_instance0=6, n_out=5)
    probe = wl.relation_messages(with_epilogue=False, free_last=False)
    inst, wit = wl.inputs(2)
    width = wl.width
    outs = np.zeros((2, wl.n_out, width), dtype=np.uint8)
    out_vals = []
    for lane in range(2):
        iv = [int.from_bytes(inst[lane, k].tobytes(), 'little') for k in range(wl.n_instance0)]
        wv = [int.from_bytes(wit[lane, k].tobytes(), 'little') for k in range(wl.n_witness)]
        run = oracle_lane(wl.mod_le, iv, wv, probe, width, trace=False)
        out_vals.append([run.get(wid) for wid in wl.output_wire_ids()])
        for t, v in enumerate(out_vals[-1]):
            outs[lane, t] = np.frombuffer(v.to_bytes(width, 'little'), dtype=np.uint8)
    wl.set_expected_outputs(inst, outs, corrupt_every=2)   # lane 0 FALSE, lane 1 TRUE
    n_pairs = {}
    for opts in ({}, {'pair': 0}, {'fuse': 0}, {'sort_by_operand': 0}, {'sort_by_operand': 1}):
        ev = zk.Evaluator()
        for k, v in opts.items():
            ev.set_option(k, str(v))
        ev.declare_inputs(wl.n_instance, wl.n_witness)
        for m in wl.relation_messages():
            ev.ingest_message(m)
        ev.finalize()
        ops, launches, consts, slot_of = ev.schedule_dump()
        info = ev.schedule_info()
        n_pairs[str(opts)] = int((((np.asarray(ops).reshape(-1, 8)[:, 1] >> 12) & 3) != 0).sum())
        for lane in range(2):
            iv = [int.from_bytes(inst[lane, k].tobytes(), 'little') for k in range(wl.n_instance)]
            wv = [int.from_bytes(wit[lane, k].tobytes(), 'little') for k in range(wl.n_witness)]
            for shuffle in (None, 3):
                slots, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'],
                                                           wl.p, iv, wv, shuffle_seed=shuffle)
                assert not noncanon
                assert (ff is None) == (lane == 1), (opts, lane, ff)
                if ff is not None:
                    assert ff == 0  # the first output comparison
    assert n_pairs['{}'] > 20 and n_pairs["{'pair': 0}"] == 0 and n_pairs["{'fuse': 0}"] == 0


def test_relation_is_split_into_100k_gate_messages():
    wl = workloads.ArithLayered(W=4096, D=50)
    msgs = wl.relation_messages()
    assert len(msgs) == 3
    ev = zk.Evaluator()
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in msgs:
        ev.ingest_message(m)
    assert ev.host_violations() == []
    assert ev.n_value_ops == 4096 * 51 + 64 * 4 and ev.n_asserts == 64


def _sim_lane(ev, p, inst, wit):
    ops, launches, consts, slot_of = ev.schedule_dump()
    info = ev.schedule_info()
    slots, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p,
                                               inst, wit, shuffle_seed=3)
    return slots, ff, slot_of, info


def test_state_persists_across_relation_messages():
    """Scope, functions and queues live across messages (evaluator.rs:158-170,273-284); the
    iterator map does not (:286).  A function defined in message 1 is called from message 2."""
    from zkinterface_ir_amd import sieve_writer as sw
    mod = bytes([101])
    inst = sw.write_instance(mod, [bytes([7]), bytes([9])])
    wit = sw.write_witness(mod, [bytes([3])])
    fn = [('sq', 1, 1, 0, 0, [('mul', 0, 1, 1)])]
    rel1 = sw.write_relation(mod, 'arithmetic', '@function', fn, [('instance', 0), ('witness', 1), ('call', 'sq', [2], [1])])
    rel2 = sw.write_relation(mod, 'arithmetic', '@function,@for', [], [
        ('instance', 3), ('call', 'sq', [4], [2]),                # 3^2 = 9, 9^2 = 81
        ('for', 'i', 0, 1, [(5, 6)], ('call', 'sq', [('add', ('name', 'i'), ('const', 5))], [('add', ('name', 'i'), ('const', 3))])),
        ('mulc', 7, 3, bytes([100])), ('add', 8, 2, 7), ('assert_zero', 8),  # 9 - instance[1] == 0
        ('free', 0, 8)])
    bufs = [inst, wit, rel1, rel2]
    ref = OracleRun(buffers=bufs)
    assert ref.violations == [] and ref.n_live_wires() == 0
    ev = zk.Evaluator.from_messages(bufs)
    assert ev.host_violations() == []
    ev.finalize(retain_all=True)
    slots, ff, slot_of, info = _sim_lane(ev, 101, [7, 9], [3])
    assert ff is None
    kinds, _, _ = ev.tape()
    vals = [program_sim.from_device_form(slots[slot_of[i]], 101, info['words_per_const'])
            for i in range(len(kinds)) if kinds[i] != 9]
    assert vals == ref.trace_values()
    # an unknown iterator name is a panic in the reference (iterators.rs:399-400): latched as such
    rel3 = sw.write_relation(mod, 'arithmetic', '@function,@for', [], [
        ('constant', 20, bytes([2])),
        ('for', 'i', 0, 0, [21], ('call', 'sq', [('add', ('name', 'j'), ('const', 21))], [('const', 20)]))])
    got = zk.Evaluator.from_messages(bufs + [rel3]).host_violations()
    ref3 = OracleRun(buffers=bufs + [rel3])
    assert ref3.panicked and got == ref3.violations and 'Unknown iterator name j' in got[0]


def test_sparse_wire_ids_and_empty_streams():
    """ids beyond the dense scope range, zero instances / witnesses, and an empty relation"""
    from zkinterface_ir_amd import sieve_writer as sw
    mod = bytes([101])
    big = (1 << 40) + 5
    rel = sw.write_relation(mod, 'arithmetic', 'simple', [], [
        ('constant', big, bytes([5])), ('constant', big + 1, bytes([96])), ('add', 2 ** 63, big, big + 1),
        ('assert_zero', 2 ** 63), ('free', big, big + 1), ('free', 2 ** 63, None)])
    ref = OracleRun(buffers=[rel])
    assert ref.violations == []
    ev = zk.Evaluator.from_messages([rel])
    assert ev.host_violations() == [] and ev.n_instance == 0 and ev.n_witness == 0
    ev.finalize()
    _, ff, _, _ = _sim_lane(ev, 101, [], [])
    assert ff is None
    empty = sw.write_relation(mod, 'arithmetic', 'simple', [], [])
    assert zk.Evaluator.from_messages([empty]).host_violations() == OracleRun(buffers=[empty]).violations == [
        'Did not receive any gate to verify.']
    # a truncated stream ends quietly (read_exact fails -> end of stream, utils.rs:27-41)
    assert zk.Evaluator.from_messages([rel[:-3]]).host_violations() == OracleRun(buffers=[rel[:-3]]).violations


def _record_through_trait(ev, p):
    """z = w0*w0 + 3 - i0 ; assert z == 0 ; also a weighted-style product assert (w1 * z)"""
    ev.backend_set_field(p.to_bytes(32, 'little'), 1, False)
    w0 = ev.backend_witness(0)
    w1 = ev.backend_witness(1)
    i0 = ev.backend_instance(0)
    sq = ev.backend_multiply(w0, w0)
    s3 = ev.backend_add_constant(sq, bytes([3]))
    neg = ev.backend_mul_constant(i0, (p - 1).to_bytes(32, 'little'))
    z = ev.backend_add(s3, neg)
    zc = ev.backend_copy(z)
    ev.backend_assert_zero(zc, 41)
    k = ev.backend_constant(bytes([7]))
    wz = ev.backend_multiply(ev.backend_multiply(w1, k), z)
    ev.backend_assert_zero(wz, 42)
    return [w0, w1, i0, sq, s3, neg, z, zc, k, wz]


def test_trait_level_entry_points_record_a_tape():
    p = circuits.BN254_R
    ev = zk.Evaluator()
    handles = _record_through_trait(ev, p)
    kinds, a, b = ev.tape()
    assert [zk.KIND_NAMES[int(k)] for k in kinds] == ['witness', 'witness', 'instance', 'mul', 'addc', 'mulc', 'add', 'copy',
                                                     'assert_zero', 'constant', 'mul', 'mul', 'assert_zero']
    assert list(ev.assert_wires()) == [41, 42]
    assert ev.n_instance == 1 and ev.n_witness == 2
    ev.finalize(retain_all=True)
    for (w0, w1, i0, fails) in [(5, 9, 28, None), (5, 9, 29, 0), (0, 0, 3, None), (1, 0, 5, 0)]:
        slots, ff, slot_of, info = _sim_lane(ev, p, [i0], [w0, w1])
        assert ff == fails
        z = (w0 * w0 + 3 - i0) % p
        assert program_sim.from_device_form(slots[slot_of[handles[6]]], p, info['words_per_const']) == z
    # misuse is reported through the status code + zkgpu_last_error, with the reference's strings where they exist
    ev2 = zk.Evaluator()
    with pytest.raises(zk.ZkGpuError, match='Modulus is not initiated'):
        ev2.backend_constant(bytes([1]))
    with pytest.raises(zk.ZkGpuError, match='Modulus cannot be zero'):
        ev2.backend_set_field(bytes([0]))
    with pytest.raises(zk.ZkGpuError, match='Field should be of degree 1'):
        ev2.backend_set_field(bytes([101]), degree=2)
    ev2.backend_set_field(bytes([101]))
    # and / xor / not over an odd field are the integer bit operations of PlaintextBackend (evaluator.rs:924-938)
    w_and = ev2.backend_and(ev2.backend_constant(bytes([1])), ev2.backend_constant(bytes([1])))
    assert zk.KIND_NAMES[int(ev2.tape()[0][w_and])] == 'and'
    ev2.backend_constant(bytes([101]))     # >= p: recorded; whether its residue will do is decided at finalize (test_unreduced.py)
