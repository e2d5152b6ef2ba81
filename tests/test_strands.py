"""Strands: consecutive narrow levels (the dependency chains of a structured relation -- one loop iteration feeding the
next) walked by one workgroup per lane block with a barrier between levels, instead of one launch per level
(zkinterface-ir_amd/csrc/device/replay_kernels.hpp replay_strand_kernel; option "strand_width").  The workload is the
reference's example shape (producers/examples.rs:72-212: For over a named function with a nested call and a Switch)
with every iteration reading the previous one's result."""
import numpy as np
import pytest

import program_sim
from helpers import oracle_lane
from test_fuzz_host import expected_product_violations
import zkinterface_ir_amd as zk
from zkinterface_ir_amd import workloads


def _session(wl, strand_width, stream='0'):
    ev = zk.Evaluator()
    ev.set_option('strand_width', str(strand_width))
    ev.set_option('stream', stream)
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in wl.relation_messages():
        ev.ingest_message(m)
    assert ev.host_violations() == []
    ev.finalize()
    return ev


def _ints(arr, lane, n):
    return [int.from_bytes(arr[lane, k].tobytes(), 'little') for k in range(n)]


@pytest.mark.parametrize('strand_width', [3, 17, 64])
def test_chained_relation_schedules_into_strands_and_computes_the_oracle_verdicts(strand_width):
    wl = workloads.StructuredArith(N=12, chained=True)
    ev = _session(wl, strand_width)
    info = ev.schedule_info()
    assert info['levels'] > 5 * wl.N                       # a chain: several levels per iteration
    if strand_width >= 17:
        assert info['launches'] < info['levels'] // 4      # ... most of them walked inside strands
    ops, launches, consts, _ = ev.schedule_dump()
    inst, wit, bad = wl.inputs(4, corrupt_every=2)
    for lane in range(4):
        iv, wv = _ints(inst, lane, wl.n_instance), _ints(wit, lane, wl.n_witness)
        ref = oracle_lane(wl.mod_le, iv, wv, wl.relation_messages(), wl.width, trace=False)
        _, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], wl.p, iv, wv)
        assert not noncanon and expected_product_violations(ev, ff) == ref.violations, lane
        assert (ref.violations == []) == (lane % 2 == 1)


def _strands_of(ev):
    ops, launches, _, _ = ev.schedule_dump()
    out = []
    for k, l in enumerate(launches):
        sl = ev.strand_levels(k)
        if sl is not None:
            out.append((ops, int(l[0]), int(l[1]), sl[0], sl[1]))
    return out


_PASSES = ('strand_prefetch', 'strand_merge', 'strand_reassociate', 'strand_split_inputs')


@pytest.mark.parametrize('strand_width,off', [(3, ()), (17, ()), (64, ())] + [(64, (p,)) for p in _PASSES] + [(64, _PASSES)])
def test_no_entry_of_a_level_touches_what_another_wave_of_it_writes(strand_width, off):
    """the strand kernel runs a level's entries on four waves at once; program_sim runs them in turn: the static rule that
    makes the two the same (program_sim.strand_hazards), and the oracle's verdicts from the simulated program, on the chained
    relation and on independent iterations, with each pass over the finished strand -- operands copied into LDS ahead of
    their readers, levels joined without a barrier, products re-associated off the chain, inputs fetched ahead of their
    conversion -- on, off alone, and all off"""
    seen = 0
    for wl in (workloads.StructuredArith(N=40, chained=True), workloads.StructuredArith(N=12, chained=False)):
        ev = zk.Evaluator()
        ev.set_option('strand_width', str(strand_width))
        for p in _PASSES:
            ev.set_option(p, '0' if p in off else '1')
        ev.declare_inputs(wl.n_instance, wl.n_witness)
        for m in wl.relation_messages():
            ev.ingest_message(m)
        ev.finalize()
        info = ev.schedule_info()
        copies = nops = raws = convs = 0
        for ops, first, count, level_ptr, lds_slots in _strands_of(ev):
            assert int(level_ptr[0]) == 0 and int(level_ptr[-1]) == count and all(np.diff(level_ptr.astype(np.int64)) > 0)
            lds = [int(x) & 0xFFFF for x in ops[first:first + count, 0] if int(x) & 0x40000000]
            assert all(k < lds_slots for k in lds)
            assert program_sim.strand_hazards(ops, first, level_ptr) == []
            for o in ops[first:first + count]:
                kind = int(o[1]) & 0xFF
                copies += kind == 5 and bool(int(o[0]) & 0x40000000) and not int(o[2]) & 0x40000000
                nops += kind == 0
                raws += kind == program_sim.OP['input_raw']
                convs += kind == program_sim.OP['input_conv']
            seen += 1
        assert raws == convs
        if wl.chained and strand_width >= 17:    # (the Switch weights of the chain are made level-wide in front of it)
            assert (copies > 0) == ('strand_prefetch' not in off), copies
            assert (raws > 0) == ('strand_split_inputs' not in off), raws
            if 'strand_merge' not in off:
                assert nops > 0
        if 'strand_merge' in off:
            assert nops == 0      # (no-ops only ever pad a joined level)
        # ... and the program computes the oracle's verdicts
        ops, launches, consts, _ = ev.schedule_dump()
        inst, wit, _bad = wl.inputs(2, corrupt_every=2)
        for lane in range(2):
            iv, wv = _ints(inst, lane, wl.n_instance), _ints(wit, lane, wl.n_witness)
            ref = oracle_lane(wl.mod_le, iv, wv, wl.relation_messages(), wl.width, trace=False)
            _, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], wl.p, iv, wv)
            assert not noncanon and expected_product_violations(ev, ff) == ref.violations, (lane, off)
    assert seen >= 1


@pytest.mark.gpu
@pytest.mark.parametrize('strand_width,stream', [(17, '0'), (3, '0'), (64, '0'), (17, '48')])
def test_chained_relation_on_gpu(strand_width, stream):
    """every lane's verdict of a 40-iteration chain (about 30,000 backend calls, 200 levels) against the closed form and,
    for sampled lanes, the oracle's violation text; ragged batch; the same with a streamed ingest"""
    wl = workloads.StructuredArith(N=40, chained=True)
    batch = 200
    inst, wit, bad = wl.inputs(batch, corrupt_every=7)
    ev = _session(wl, strand_width, stream)
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (batch - bad, bad)
    first, flags = ev.lane_results(batch)
    assert np.array_equal(first == zk.NO_FAIL, np.arange(batch) % 7 != 0) and not flags.any()
    for lane in (0, 1, 7, 63, 64, 199):
        ref = oracle_lane(wl.mod_le, _ints(inst, lane, wl.n_instance), _ints(wit, lane, wl.n_witness), wl.relation_messages(),
                          wl.width, trace=False)
        assert ev.get_violations(lane) == ref.violations, lane
    # a second replay of the same inputs: the strands leave the table in a state that replays identically
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (batch - bad, bad)


@pytest.mark.gpu
def test_independent_iterations_on_gpu_against_the_oracle():
    """the bench's structured workload at a small size: For / Call / Switch with independent iterations (wide levels)"""
    wl = workloads.StructuredArith(N=48)
    batch = 130
    inst, wit, bad = wl.inputs(batch, corrupt_every=5)
    ev = _session(wl, 17)
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (batch - bad, bad)
    for lane in (0, 1, 5, 64, 129):
        ref = oracle_lane(wl.mod_le, _ints(inst, lane, wl.n_instance), _ints(wit, lane, wl.n_witness), wl.relation_messages(),
                          wl.width, trace=False)
        assert ev.get_violations(lane) == ref.violations, lane
