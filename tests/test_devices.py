"""One process driving several GPUs (option "devices"; SURVEY.md 8b threading row, 8e): the lanes of a batch split
over several engines, each on a host thread of its own, counts combined.  A one-GPU box lists its device several times
("0,0,0"): same lane split, same threads, counts summed on the host (RCCL needs distinct devices and is used when the
list has them -- that branch cannot run here).  Everything a caller can ask for must equal the single-engine answer."""
import os
import subprocess

import numpy as np
import pytest

import zkinterface_ir_amd as zk
from helpers import REF_EXAMPLES, ROOT, batch_arrays, oracle_lane
from zkinterface_ir_amd import workloads


def test_devices_option_is_parsed_on_the_host():
    ev = zk.Evaluator()
    ev.set_option('devices', '0,1,2,3')
    assert ev.n_engines == 4
    ev.set_option('devices', '')
    assert ev.n_engines == 1
    with pytest.raises(zk.ZkGpuError, match='comma-separated list'):
        ev.set_option('devices', '0,x')


def _session(wl, batch, devices, inst, wit):
    ev = zk.Evaluator()
    if devices:
        ev.set_option('devices', devices)
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in wl.relation_messages():
        ev.ingest_message(m)
    ev.finalize()
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    ev.replay()
    ev.synchronize()
    return ev


@pytest.mark.gpu
@pytest.mark.parametrize('batch,devices', [(700, '0,0,0'), (64, '0,0,0'), (130, '0,0'), (1000, '0,0,0,0,0')])
def test_lane_split_over_engines_gives_the_single_engine_answers(batch, devices):
    wl = workloads.StructuredArith(N=24)         # switch, calls, loops: violations name wires
    inst, wit, bad = wl.inputs(batch, corrupt_every=7)
    one = _session(wl, batch, None, inst, wit)
    many = _session(wl, batch, devices, inst, wit)
    assert many.n_engines == devices.count(',') + 1 and one.n_engines == 1
    assert many.counts() == one.counts() == (batch - bad, bad)
    f1, g1 = one.lane_results(batch)
    f2, g2 = many.lane_results(batch)
    assert np.array_equal(f1, f2) and np.array_equal(g1, g2)
    for lane in (0, 1, 7, batch // 2, batch - 1):
        assert many.get_violations(lane) == one.get_violations(lane)
    # a second batch through the same engines, different size
    b2 = max(1, batch // 3)
    many.set_inputs(inst[:b2].tobytes(), wit[:b2].tobytes(), b2)
    many.replay()
    many.synchronize()
    assert many.counts() == (b2 - sum(1 for i in range(b2) if i % 7 == 0), sum(1 for i in range(b2) if i % 7 == 0))
    with pytest.raises(zk.ZkGpuError, match='not available with several devices'):
        many.replay_timed()


@pytest.mark.gpu
def test_wire_values_are_gathered_in_lane_order():
    wl = workloads.ArithLayered(W=128, D=5, n_instance0=8, n_out=4)
    batch = 300
    inst, wit = wl.inputs(batch)
    msgs = wl.relation_messages(with_epilogue=False, free_last=False)
    cols = {}
    for devices in (None, '0,0,0'):
        ev = zk.Evaluator()
        if devices:
            ev.set_option('devices', devices)
        ev.declare_inputs(wl.n_instance0, wl.n_witness)
        for m in msgs:
            ev.ingest_message(m)
        ev.finalize()
        ev.set_inputs(np.ascontiguousarray(inst[:, :wl.n_instance0]).tobytes(), wit.tobytes(), batch)
        ev.replay()
        ev.synchronize()
        cols[devices] = [ev.get(w, batch) for w in wl.output_wire_ids()]
    assert cols[None] == cols['0,0,0']
    lane = 200
    iv = [int.from_bytes(inst[lane, k].tobytes(), 'little') for k in range(wl.n_instance0)]
    wv = [int.from_bytes(wit[lane, k].tobytes(), 'little') for k in range(wl.n_witness)]
    ref = oracle_lane(wl.mod_le, iv, wv, msgs, wl.width, trace=False)
    assert [c[lane] for c in cols['0,0,0']] == [ref.get(w) for w in wl.output_wire_ids()]


@pytest.mark.gpu
def test_c_example_splits_a_batch_over_the_listed_devices(tmp_path):
    exe = str(tmp_path / 'evaluate_batch_devices')
    subprocess.check_call(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'evaluate_batch_devices.c'),
                           '-L', os.path.dirname(zk.LIB_PATH), '-lzkgpu',
                           '-Wl,-rpath,' + os.path.dirname(zk.LIB_PATH), '-o', exe])
    r = subprocess.run([exe, '--devices', '0,0', '--batch', '200'] + REF_EXAMPLES, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert '200 statements over 2 engine(s) [devices 0,0]: 199 TRUE, 1 FALSE' in r.stdout
    assert 'lane 0: TRUE' in r.stdout and 'lane 1: FALSE: Wire_' in r.stdout
    r = subprocess.run([exe, '--batch', '64'] + REF_EXAMPLES, capture_output=True, text=True)   # every visible GPU
    assert r.returncode == 0 and '63 TRUE, 1 FALSE' in r.stdout


@pytest.mark.gpu
def test_counts_through_rccl_with_a_one_device_communicator():
    """option "force_rccl": zkgpu_counts runs the in-library RCCL reduction -- dlopen of librccl, ncclCommInitAll,
    ncclGroupStart / ncclAllReduce(2 x u64, sum) on the engine's stream / ncclGroupEnd -- with a communicator of one rank,
    so that the code path of a multi-GPU `devices` list has run against the real RCCL on a one-GPU box.  The counts must
    be the plain ones, batch after batch, and the session says that RCCL answered."""
    wl = workloads.StructuredArith(N=24)
    batch = 500
    inst, wit, bad = wl.inputs(batch, corrupt_every=7)
    plain = _session(wl, batch, None, inst, wit)
    assert plain.counts() == (batch - bad, bad) and plain.rccl_reductions == 0
    ev = zk.Evaluator()
    ev.set_option('force_rccl', '1')
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in wl.relation_messages():
        ev.ingest_message(m)
    ev.finalize()
    for k, b in enumerate((batch, 130, batch)):
        ev.set_inputs(inst[:b].tobytes(), wit[:b].tobytes(), b)
        ev.replay()
        ev.synchronize()
        nb = sum(1 for i in range(b) if i % 7 == 0)
        assert ev.counts() == (b - nb, nb)
        assert ev.rccl_reductions == k + 1 and ev.rccl_note() == ''
    # per-lane results are untouched by the way the counts were combined
    assert np.array_equal(ev.lane_results(batch)[0], plain.lane_results(batch)[0])
    # a device listed twice cannot be an RCCL communicator: with force_rccl that is an error, not a silent host sum
    two = zk.Evaluator()
    two.set_option('devices', '0,0')
    two.set_option('force_rccl', '1')
    two.declare_inputs(wl.n_instance, wl.n_witness)
    for m in wl.relation_messages():
        two.ingest_message(m)
    two.finalize()
    two.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    two.replay()
    two.synchronize()
    with pytest.raises(zk.ZkGpuError, match='force_rccl'):
        two.counts()


@pytest.mark.gpu
@pytest.mark.parametrize('retain', [False, True])
def test_field_segments_over_several_engines(retain):
    """a relation whose field characteristic changes between messages (tests/test_field_segments.py) with the lanes split
    over three engines: every engine runs a chain of its own (one engine per field segment); counts, per-lane results,
    violation texts and the wire values of every call equal the single-engine session's and the oracle's"""
    import test_field_segments as fs
    from zkinterface_ir_amd import sieve_writer as sw
    msgs = fs._messages(fs.GROW)
    base = fs._lanes_grow()
    rows = [base[k % len(base)] for k in range(200)]
    sessions = {}
    for devices in (None, '0,0,0'):
        ev = zk.Evaluator()
        if devices:
            ev.set_option('devices', devices)
        ev.declare_inputs(2, 2)
        for m in msgs:
            ev.ingest_message(m)
        ev.finalize(retain_all=retain)
        inst, wit = batch_arrays([r[0] for r in rows], [r[1] for r in rows], ev.elem_bytes)
        ev.set_inputs(inst, wit, len(rows))
        ev.replay()
        ev.synchronize()
        sessions[devices] = ev
    one, many = sessions[None], sessions['0,0,0']
    assert many.n_engines == 3 and many.n_field_segments == 2
    assert many.counts() == one.counts() == (150, 50)
    assert np.array_equal(one.lane_results(len(rows))[0], many.lane_results(len(rows))[0])
    for lane in (0, 1, 2, 3, 66, 130, 199):
        ref = oracle_lane(sw.int_to_le(fs.P1), rows[lane][0], rows[lane][1], msgs, 32, trace=False)
        assert many.get_violations(lane) == one.get_violations(lane) == ref.violations, lane
    if retain:
        assert many.dump_trace_values(len(rows)) == one.dump_trace_values(len(rows))


@pytest.mark.gpu
def test_r1cs_rows_over_two_engines_give_the_single_engine_results():
    """the C5 shape, small: witness generation level by level and the row check with the lanes split over two engines
    (rows replicated like the program): first failing rows, counts and the values of sampled variables equal the
    single-engine session's"""
    wl = workloads.R1csSynthetic(M=600, n_base=32, n_coefs=50, seed=5)
    batch = 200
    row_ptr, tv, tc, cb = wl.csr()
    results = {}
    for devices in (None, '0,0'):
        ev = zk.Evaluator()
        if devices:
            ev.set_option('devices', devices)
        ev.declare_inputs(0, wl.n_witness)
        ev.ingest_message(wl.base_relation())
        ev.finalize(retain_all=True)
        ev.r1cs_load_csr(row_ptr, tv, tc, cb, wl.width, wl.M)
        w = wl.witnesses(batch)
        ev.set_inputs(None, w.tobytes(), batch)
        ev.replay()
        lo = 0
        for hi in wl.level_bounds:
            ev.r1cs_assign(lo, int(hi) - lo)
            lo = int(hi)
        zl = ev.r1cs_get_var(wl.last_z, batch)
        for lane in range(batch):
            v = zl[lane] if lane % 9 else (zl[lane] + 1) % wl.p
            w[lane, wl.n_base] = np.frombuffer(v.to_bytes(wl.width, 'little'), dtype=np.uint8)
        ev.set_inputs(None, w.tobytes(), batch)
        ev.replay()
        lo = 0
        for hi in wl.level_bounds:
            ev.r1cs_assign(lo, int(hi) - lo)
            lo = int(hi)
        ev.r1cs_check()
        ff, counts = ev.r1cs_results(batch)
        results[devices] = (np.asarray(ff).copy(), tuple(counts), zl, ev.r1cs_get_vars([wl.n_base + 1, wl.n_base + 1 + wl.M // 2, wl.last_z], batch))
    one, two = results[None], results['0,0']
    n_bad = sum(1 for lane in range(batch) if lane % 9 == 0)
    assert one[1] == two[1] == (batch - n_bad, n_bad)
    assert np.array_equal(one[0], two[0]) and one[2] == two[2] and one[3] == two[3]
