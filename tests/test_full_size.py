"""Full-size parity of BASELINE configs[3] (C4: GF(2), 10.5M gates, batch 4096) and configs[4] (C5: 2^20-row R1CS over
BN254, batch 1024), pinned the way C2 is: committed digests made by the oracle / by Python integers
(tests/golden/c4_digests.json, c5_digests.json + their generators), a CPU checker for every lane that shares nothing
with the product but the workload definition (tests/cpu_checkers.py), and the known satisfied count.  No expected
value in this file comes from the GPU.

CPU tier: the checkers themselves against the oracle (small instances) and against the committed digests (full size).
GPU tier: both GF(2) kernels and the R1CS assign / check kernels against all of it, through the C ABI."""
import json
import os

import numpy as np
import pytest

import cpu_checkers
import zkinterface_ir_amd as zk
from helpers import ROOT, oracle_lane
from zkinterface_ir_amd import workloads

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def _fixture(name):
    return json.load(open(os.path.join(GOLDEN, name)))


# ---------------------------------------------------------------- CPU tier: the checkers are pinned first
def test_bool_checker_agrees_with_the_oracle_on_every_lane_of_a_small_relation():
    wl = workloads.BoolLayered(W=128, D=10, n_instance0=16, n_out=8)
    batch = 70   # two 64-witness words, the second ragged
    inst, wit = wl.inputs(batch)
    outs = cpu_checkers.bool_layered_outputs(wl, inst, wit)
    msgs = wl.relation_messages(with_epilogue=False, free_last=False)
    for lane in range(batch):
        ref = oracle_lane(wl.mod_le, inst[lane, :wl.n_instance0, 0].tolist(), wit[lane, :, 0].tolist(), msgs, 1, trace=False)
        assert [ref.get(w) for w in wl.output_wire_ids()] == outs[lane].tolist(), lane


@pytest.fixture(scope='module')
def c4_reference():
    """all 4096 lanes x 64 outputs of the full-size C4 relation from the numpy checker, pinned to the oracle digests"""
    fx = _fixture('c4_digests.json')
    wl = workloads.BoolLayered()
    inst, wit = wl.inputs(4096)
    outs = cpu_checkers.bool_layered_outputs(wl, inst, wit)
    for lane, want in fx['lanes'].items():
        assert cpu_checkers.bool_digest(outs[int(lane)]) == want['sha256'], lane
        assert ''.join(str(int(b)) for b in outs[int(lane)]) == want['bits']
    # every lane against the committed per-lane hashes (tests/golden/make_all_lanes.py)
    cpu_checkers.check_against_all_lanes_fixture(_fixture('c4_all_lanes.json'), outs)
    return wl, inst, wit, outs


def test_full_size_c4_checker_reproduces_the_committed_oracle_digests(c4_reference):
    wl, _, _, outs = c4_reference
    assert outs.shape == (4096, 64) and 0.3 < outs.mean() < 0.7      # not a degenerate circuit


def test_r1cs_python_checker_agrees_with_the_c_row_check():
    """the Python-integer assignment (generator of c5_digests.json) and oracle/cpu_opt.cpp's row check are independent
    statements of the same definition: the C check accepts exactly the Python z_last as expected value"""
    from oracle_lib import r1cs_check
    wl = workloads.R1csSynthetic(M=500, n_base=16, n_coefs=40, seed=11)
    batch = 5
    w = wl.witnesses(batch)
    row_ptr, tv, tc, cb = wl.csr()
    for lane in range(batch):
        val = cpu_checkers.r1cs_lane_assignment(wl, w[lane])
        e = val[wl.last_z] if lane != 3 else (val[wl.last_z] + 1) % wl.p
        w[lane, wl.n_base] = np.frombuffer(e.to_bytes(wl.width, 'little'), dtype=np.uint8)
    ff, _ = r1cs_check(row_ptr, tv, tc, cb, wl.mod_le, w, wl.n_base + 1 + wl.M, wl.M, 2)
    assert ff.tolist() == [0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, wl.M, 0xFFFFFFFF]


def test_c5_fixture_is_the_python_integer_assignment_of_its_first_lane():
    """one lane of tests/golden/c5_digests.json regenerated here (about 10 s): the fixture is what its generator says"""
    fx = _fixture('c5_digests.json')
    wl = workloads.R1csSynthetic()
    w = wl.witnesses(1024)
    val = cpu_checkers.r1cs_lane_assignment(wl, w[1])
    sample = cpu_checkers.r1cs_sample_vars(wl)
    assert len(sample) == fx['sampled_variables']
    assert cpu_checkers.r1cs_digest([val[v] for v in sample], wl.width) == fx['lanes']['1']['sha256']
    assert str(val[wl.last_z]) == fx['lanes']['1']['last_z']


# ---------------------------------------------------------------- GPU tier
@pytest.mark.gpu
@pytest.mark.parametrize('path', ['lds', 'hbm'])
def test_full_size_c4_against_the_cpu_checker_and_the_oracle_digests(c4_reference, path):
    """BASELINE configs[3] on one GF(2) kernel: (1) the 64 output wires of ALL 4096 lanes equal the CPU checker's
    (which the fixture pins to the oracle on six lanes), (2) with the epilogue and every 97th statement damaged, the
    satisfied count, every lane's first failing assert and the violation text are the reference's."""
    wl, inst, wit, outs = c4_reference
    batch = 4096
    probe = zk.Evaluator()
    probe.set_option('bool_path', path)
    probe.declare_inputs(wl.n_instance0, wl.n_witness)
    for m in wl.relation_messages(with_epilogue=False, free_last=False):
        probe.ingest_message(m)
    probe.finalize()
    probe.set_inputs(np.ascontiguousarray(inst[:, :wl.n_instance0]).tobytes(), wit.tobytes(), batch)
    assert probe.uses_lds_path() == (path == 'lds')
    probe.replay()
    probe.synchronize()
    got = np.zeros((batch, wl.n_out), dtype=np.uint8)
    for t, wid in enumerate(wl.output_wire_ids()):
        got[:, t] = probe.get(wid, batch)
    assert np.array_equal(got, outs)
    probe.close()

    inst = inst.copy()
    n_bad = wl.set_expected_outputs(inst, outs)          # expected outputs from the CPU checker, not from the GPU
    ev = zk.Evaluator()
    ev.set_option('bool_path', path)
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in wl.relation_messages():
        ev.ingest_message(m)
    assert ev.host_violations() == []
    ev.finalize()
    info = ev.schedule_info()
    # inputs + gates + per output {Instance, Xor, the copy AssertZero makes of its wire (evaluator.rs:340)}
    assert ev.n_value_ops == wl.W * (wl.D + 1) + 3 * wl.n_out and ev.n_asserts == wl.n_out
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    assert ev.uses_lds_path() == (path == 'lds')
    if path == 'lds':
        assert (info['slots'] + 34) * 4 <= 160 * 1024     # the whole wire table of a 32-witness slice (+ scratch and
                                                          # constant slots) in one CU's LDS
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (workloads.expected_satisfied(batch), n_bad) == (4053, 43)
    first, flags = ev.lane_results(batch)
    want = np.where(np.arange(batch) % 97 == 0, 0, zk.NO_FAIL).astype(np.uint32)   # output 0 is the damaged one: assert #0
    assert np.array_equal(first, want) and not flags.any()
    assert ev.get_violations(97) == ['Wire_%d (may be weighted) should be 0, while it is not' % ((wl.D + 1) * wl.W + 1)]
    assert ev.get_violations(98) == []
    # a different output damaged on one lane: its own assert is the first to fail
    inst[98, wl.n_instance0 + 5, 0] ^= 1
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (4052, 44) and int(ev.lane_results(batch)[0][98]) == 5


@pytest.mark.gpu
def test_full_size_c4_streamed_is_the_program_scheduled_at_finalize(c4_reference):
    """BASELINE configs[3] ingested as a stream (106 messages; windows of 131072 recorded calls = 8 layers, cut at the
    level seams and scheduled by the worker thread while the next messages are parsed): levels, slots and the bytes of the
    LDS-resident kernel's program are those of the schedule made at finalize, and the statement's counts, first failing
    asserts and violation texts are the reference's for all 4096 lanes."""
    wl, inst, wit, outs = c4_reference
    batch = 4096
    inst = inst.copy()
    n_bad = wl.set_expected_outputs(inst, outs)
    msgs = wl.relation_messages()
    sessions = {}
    for stream in (0, 1):
        ev = zk.Evaluator()
        ev.set_option('bool_path', 'lds')
        ev.set_option('stream', str(stream))
        ev.declare_inputs(wl.n_instance, wl.n_witness)
        for m in msgs:
            ev.ingest_message(m)
        ev.finalize()
        sessions[stream] = ev
    sinfo = sessions[1].stream_info()
    assert sinfo['windows'] >= 80 and sinfo['streamed_windows'] == sinfo['windows']
    assert sessions[0].schedule_info() == sessions[1].schedule_info()
    P, Q = sessions[0].lds_program(0), sessions[1].lds_program(0)
    for k in ('ops8', 'rows', 'blocks', 'chunks'):
        assert np.array_equal(P[k], Q[k]), k
    sessions[0].close()
    ev = sessions[1]
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    assert ev.uses_lds_path()
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (workloads.expected_satisfied(batch), n_bad) == (4053, 43)
    first, flags = ev.lane_results(batch)
    assert np.array_equal(first, np.where(np.arange(batch) % 97 == 0, 0, zk.NO_FAIL).astype(np.uint32)) and not flags.any()
    assert ev.get_violations(97) == ['Wire_%d (may be weighted) should be 0, while it is not' % ((wl.D + 1) * wl.W + 1)]


@pytest.mark.gpu
@pytest.mark.parametrize('block_rows', [None, 4, 6, 8, 9, 10, 12])
def test_lds_kernel_block_shapes_against_the_cpu_checker(block_rows, monkeypatch):
    """The LDS-resident kernel has one instantiation per block size and, inside it, one code path per number of rows of
    a block and per position of the and -> xor change in a full block; the engine picks the block size per program.
    A relation with 11 rows per level (18,432 gates: 5 and-rows, 5 xor-rows, 1 row of nots) run with every block size
    forced (ZKGPU_LDS_BLOCK_ROWS, a tuning switch of the engine) walks levels of 1 to 3 blocks, full and short, with the
    kind change in different rows: all output wires of all lanes against the CPU checker, ragged batch."""
    if block_rows is None:
        monkeypatch.delenv('ZKGPU_LDS_BLOCK_ROWS', raising=False)
    else:
        monkeypatch.setenv('ZKGPU_LDS_BLOCK_ROWS', str(block_rows))
    wl = workloads.BoolLayered(W=18432, D=5, n_instance0=64, n_out=48, seed=0xB10C + (block_rows or 0))
    batch = 75
    inst, wit = wl.inputs(batch)
    outs = cpu_checkers.bool_layered_outputs(wl, inst, wit)
    assert 0.15 < outs.mean() < 0.85
    inst = inst.copy()
    n_bad = wl.set_expected_outputs(inst, outs, corrupt_every=7)
    ev = zk.Evaluator()
    ev.set_option('bool_path', 'lds')
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in wl.relation_messages():
        ev.ingest_message(m)
    ev.finalize()
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    assert ev.uses_lds_path()
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (batch - n_bad, n_bad)
    first, flags = ev.lane_results(batch)
    assert np.array_equal(first, np.where(np.arange(batch) % 7 == 0, 0, zk.NO_FAIL).astype(np.uint32)) and not flags.any()
    # every output bit, not only "all equal": flip one expected bit per lane and see exactly that assert fail
    for lane in range(batch):
        inst[lane, wl.n_instance0, 0] = outs[lane, 0]                       # undo the damage of set_expected_outputs
        inst[lane, wl.n_instance0 + lane % wl.n_out, 0] ^= 1
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (0, batch)
    assert ev.lane_results(batch)[0].tolist() == [lane % wl.n_out for lane in range(batch)]


@pytest.mark.gpu
@pytest.mark.parametrize('path', ['lds', 'hbm'])
def test_sha256_compression_against_hashlib_and_the_oracle(path):
    """workloads.Sha256Compress: the SHA-256 compression function as a GF(2) relation (121,072 gates, 3,920 levels of a few
    dozen gates: the narrow-level path of the LDS-resident kernel).  Every lane hashes its own message; the claimed digests
    are hashlib's, one bit flipped on every 7th lane: counts, every lane's first failing assert, the violation text and --
    for three lanes -- the oracle's run of the same messages."""
    wl = workloads.Sha256Compress()
    batch = 300
    inst, wit, bad = wl.inputs(batch, corrupt_every=7)
    msgs = wl.relation_messages()
    ev = zk.Evaluator()
    ev.set_option('bool_path', path)
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in msgs:
        ev.ingest_message(m)
    ev.finalize()
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    assert ev.uses_lds_path() == (path == 'lds')
    ev.replay()
    ev.synchronize()
    assert ev.counts() == (batch - bad, bad)
    first, flags = ev.lane_results(batch)
    want = np.where(np.arange(batch) % 7 == 0, np.arange(batch) % 256, zk.NO_FAIL).astype(np.uint32)   # assert k checks digest bit k
    assert np.array_equal(first, want) and not flags.any()
    for lane in (0, 1, 7):
        ref = oracle_lane(wl.mod_le, [int(x) for x in inst[lane, :, 0]], [int(x) for x in wit[lane, :, 0]], msgs, 1, trace=False)
        assert ev.get_violations(lane) == ref.violations, lane
    # the digest wires themselves (a session without the epilogue): hashlib's bits
    # (nothing is freed, so every wire stays readable: 121,586 of them, the HBM-table kernel)
    probe = zk.Evaluator()
    probe.set_option('bool_path', 'hbm')
    probe.declare_inputs(0, wl.n_witness)
    for m in wl.relation_messages(with_epilogue=False, free_last=False):
        probe.ingest_message(m)
    probe.finalize()
    probe.set_inputs(None, wit[:40].tobytes(), 40)
    probe.replay()
    probe.synchronize()
    got = np.stack([np.asarray(probe.get(w, 40), dtype=np.uint8) for w in wl.output_wire_ids()], axis=1)
    clean, _, _ = wl.inputs(40, corrupt_every=0)
    assert np.array_equal(got, clean[:, :, 0])


@pytest.mark.gpu
def test_lds_kernel_fuzz_of_shapes_mixes_and_block_sizes():
    """tools/fuzz_bool_lds.py, bounded: 20 random shapes of the LDS-resident GF(2) kernel -- widths from 96 to 19,000 gates
    per level, every gate mix (all and, no and, nothing but not), ragged batches, block sizes forced and free, every third
    relation ingested as a stream -- every lane's first failing assert against the numpy checker"""
    import importlib.util
    import os
    from helpers import ROOT
    spec = importlib.util.spec_from_file_location('fuzz_bool_lds', os.path.join(ROOT, 'tools', 'fuzz_bool_lds.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(20, seed=4, verbose=False) == 20


@pytest.mark.gpu
def test_full_size_c5_against_python_integer_digests_and_the_cpu_row_check():
    """BASELINE configs[4]: witness generation by zkgpu_r1cs_assign level by level, then the row check.  The generated
    variables of four lanes equal the committed Python-integer values (257 variables spread over all levels); with
    every 97th lane's expected value damaged the satisfied count is the known one and the first failing row of a
    128-lane sample (all damaged lanes included) equals the CPU row check's, which assigns from the base witness
    itself."""
    from oracle_lib import r1cs_check
    fx = _fixture('c5_digests.json')
    wl = workloads.R1csSynthetic()
    batch = 1024
    ev = zk.Evaluator()
    ev.declare_inputs(0, wl.n_witness)
    ev.ingest_message(wl.base_relation())
    ev.finalize(retain_all=True)
    row_ptr, tv, tc, cb = wl.csr()
    ev.r1cs_load_csr(row_ptr, tv, tc, cb, wl.width, wl.M)
    w = wl.witnesses(batch)
    ev.set_inputs(None, w.tobytes(), batch)
    ev.replay()
    lo = 0
    for hi in wl.level_bounds:
        ev.r1cs_assign(lo, int(hi) - lo)
        lo = int(hi)
    assert lo == wl.M
    sample = cpu_checkers.r1cs_sample_vars(wl)
    vals = ev.r1cs_get_vars(sample, batch)
    for lane, want in fx['lanes'].items():
        assert cpu_checkers.r1cs_digest(vals[int(lane)], wl.width) == want['sha256'], lane
        assert str(vals[int(lane)][-1]) == want['last_z']
    # expected value E := z_last, damaged on every 97th lane; a fifth digest lane keeps its fixture value
    bad = 0
    for lane in range(batch):
        v = vals[lane][-1]
        if lane % 97 == 0:
            v = (v + 1) % wl.p
            bad += 1
        w[lane, wl.n_base] = np.frombuffer(v.to_bytes(wl.width, 'little'), dtype=np.uint8)
    ev.set_inputs(None, w.tobytes(), batch)
    ev.replay()        # reloads the base variables; the assigned z stay in their slots behind the program's
    ev.r1cs_check()
    ff, counts = ev.r1cs_results(batch)
    assert counts == (workloads.expected_satisfied(batch), bad) == (1013, 11)
    want = np.where(np.arange(batch) % 97 == 0, wl.M, zk.NO_FAIL).astype(np.uint32)   # the comparison row is row M
    assert np.array_equal(ff, want)
    lanes = sorted(set(range(0, batch, 97)) | set(range(64)) | {int(k) for k in fx['lanes']} | set(range(960, 1024)))[:160]
    ff_cpu, _ = r1cs_check(row_ptr, tv, tc, cb, wl.mod_le, np.ascontiguousarray(w[lanes]), wl.n_base + 1 + wl.M, wl.M, 16)
    assert np.array_equal(ff_cpu, ff[lanes])
    # a damaged product-row variable is found at its own row: overwrite z of row 12345 on lane 7 through its definition
    # (the row check reads what assign wrote, so damage the base witness instead and re-check WITHOUT re-assigning)
    w2 = w.copy()
    w2[7, 0, 0] ^= 1
    ev.set_inputs(None, w2.tobytes(), batch)
    ev.replay()
    ev.r1cs_check()
    ff2, counts2 = ev.r1cs_results(batch)
    first_reader = int(np.where((wl.picks == 0).any(axis=1))[0][0])      # first row that reads base variable 0
    assert int(ff2[7]) == first_reader and counts2 == (1013 - 1, bad + 1)
    assert np.array_equal(np.delete(ff2, 7), np.delete(ff, 7))
