"""Streaming ingest (rust/src/consumers/evaluator.rs:286-301: the reference consumes a relation as a
stream of <= 100k-gate messages).  With option "stream" the tape is cut into windows while it is recorded and a worker
thread schedules every window as soon as it is complete; what a window may fuse, elide or recycle rests on the drop
records of the wires.  CPU tier: the streamed program interpreted by program_sim against the oracle, and program
identity however the relation is split into messages.  GPU tier: the same through the kernels."""
import numpy as np
import pytest

import circuits
import program_sim
from helpers import batch_arrays, oracle_lane
from random_circuits import Gen
from test_fuzz_host import FIELDS, expected_product_violations
import zkinterface_ir_amd as zk
from zkinterface_ir_amd import sieve_writer as sw
from zkinterface_ir_amd import workloads


def _streamed(rel_msgs, n_inst, n_wit, window, **options):
    ev = zk.Evaluator()
    ev.set_option('stream', str(window))
    for k, v in options.items():
        ev.set_option(k, str(v))
    ev.declare_inputs(n_inst, n_wit)
    for m in rel_msgs:
        ev.ingest_message(m)
    return ev


@pytest.mark.parametrize('seed', range(40))
def test_streamed_random_relations_against_oracle(seed):
    """functions / for / switch / frees cut into windows of 24 recorded calls: a value that a later window may still
    read keeps its slot, everything else is fused / elided / recycled as in the one-window schedule; every lane's
    verdict and violation text is the oracle's, and a level never overwrites what it reads (program_sim asserts)."""
    p, boolean = FIELDS[seed % len(FIELDS)]
    g = Gen(seed, p, boolean)
    rel, mod_le = g.relation(n_top=14)
    rows_i, rows_w = g.lane_inputs(3, seed + 1000)
    ev = _streamed([rel], g.n_inst, g.n_wit, 24)
    if not ev.n_value_ops and ev.host_violations():
        return
    ev.finalize()
    info, sinfo = ev.schedule_info(), ev.stream_info()
    # (a cut waits for the end of a Switch ladder, and for an entry that opens a new dependency level or twice the window)
    assert sinfo['windows'] >= 2 or len(ev.tape()[0]) < 48 + 400
    assert sinfo['streamed_windows'] == sinfo['windows']              # all of them went through the worker thread, GF(2) included
    ops, launches, consts, slot_of = ev.schedule_dump()
    for lane in range(3):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32, trace=False)
        _, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p,
                                               rows_i[lane], rows_w[lane], shuffle_seed=seed)
        assert not noncanon
        assert expected_product_violations(ev, ff) == ref.violations, (seed, lane)
    # the one-window schedule of the same relation is at most as large, and uses at most as many slots
    whole = zk.Evaluator()
    whole.declare_inputs(g.n_inst, g.n_wit)
    whole.ingest_message(rel)
    whole.finalize()
    assert whole.schedule_info()['device_ops'] <= info['device_ops']


def test_program_does_not_depend_on_how_the_relation_is_split_into_messages():
    """The cuts are a property of the tape (in front of the first entry that opens a new dependency level once `stream`
    calls are recorded, at twice that at the latest, put off while a Switch ladder is open), so 3 messages, one
    concatenated buffer, and a non-streamed ingest scheduled with the same windows at finalize give the same program,
    entry by entry."""
    wl = workloads.ArithLayered(W=1024, D=230, n_instance0=16, n_out=8)
    msgs = wl.relation_messages()
    assert len(msgs) == 3
    dumps = []
    for parts in (msgs, [b''.join(msgs)]):
        ev = _streamed(parts, wl.n_instance, wl.n_witness, 40000)
        ev.finalize()
        n_tape = len(ev.tape()[0])
        assert n_tape // 80000 + 1 <= ev.stream_info()['windows'] <= n_tape // 40000 + 1
        assert ev.stream_info()['streamed_windows'] == ev.stream_info()['windows']
        dumps.append(ev.schedule_dump())
    for x, y in zip(*dumps):
        assert np.array_equal(x, y)
    # scheduled at finalize with retain_all the stream is ignored: every value keeps its own slot
    ev = _streamed(msgs, wl.n_instance, wl.n_witness, 40000)
    ev.finalize(retain_all=True)
    assert ev.stream_info()['windows'] == 1 and ev.schedule_info()['slots'] == ev.n_value_ops
    # and the streamed program computes what the oracle computes
    ev = _streamed(msgs, wl.n_instance, wl.n_witness, 40000)
    ev.finalize()
    ops, launches, consts, _ = ev.schedule_dump()
    info = ev.schedule_info()
    inst, wit = wl.inputs(1)
    iv = [int.from_bytes(inst[0, k].tobytes(), 'little') for k in range(wl.n_instance)]
    wv = [int.from_bytes(wit[0, k].tobytes(), 'little') for k in range(wl.n_witness)]
    ref = oracle_lane(wl.mod_le, iv, wv, msgs, wl.width, trace=False)
    _, ff, _ = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], wl.p, iv, wv)
    assert expected_product_violations(ev, ff) == ref.violations and ff is not None   # expected outputs are all 0 here
    assert info['slots'] < 4 * wl.W        # slots are recycled across windows (three live layers at most)


def test_switch_ladders_are_never_cut_and_still_become_one_entry():
    """a window boundary that falls inside the 352-call exponent ladder of a Switch weight is put off to the ladder's
    end, so the Fermat rewrite (one `x != 0` entry) works window by window"""
    _i, _w, rel = circuits.arith_example(circuits.BN254_R)
    sizes = {}
    for window in (0, 100):
        ev = zk.Evaluator()
        ev.set_option('stream', str(window))
        ev.declare_inputs(3, 4)
        ev.ingest_message(rel)
        ev.finalize()
        sizes[window] = int(((ev.schedule_dump()[0][:, 1] & 0xFF) != 0).sum())    # entries that do something
        if window:
            assert ev.stream_info()['windows'] >= 3
    # 965 recorded calls either way (a strand adds entries for itself: copies of wire-table operands into LDS ahead of
    # their readers; the no-ops that keep a chain on one wave are not counted)
    assert sizes[100] < 200 and sizes[0] < 120


def test_trait_level_recording_with_drops_streams_too():
    """a caller that drives the ZKBackend entry points itself (the Rust Evaluator of INTEGRATION.md) reports dropped
    wires with zkgpu_backend_drop (`impl Drop` of its Wire type); without drops every value stays materialised until
    finalize, with them the windows fuse and recycle"""
    p = circuits.BN254_R
    results = {}
    for drops in (False, True):
        ev = zk.Evaluator()
        ev.set_option('stream', '64')
        ev.backend_set_field(p.to_bytes(32, 'little'))
        x = ev.backend_witness(0)
        acc = ev.backend_witness(1)
        for k in range(400):                   # acc = (acc * x + acc) chained: every intermediate has one reader
            t = ev.backend_multiply(acc, x)
            nxt = ev.backend_add(t, acc)
            if drops:
                ev.backend_drop(t)
                ev.backend_drop(acc)
            acc = nxt
        out = ev.backend_add_constant(acc, (p - 5).to_bytes(32, 'little'))
        ev.backend_assert_zero(out, 7)
        ev.finalize()
        info = ev.schedule_info()
        ops, launches, consts, _ = ev.schedule_dump()
        want = 3
        for k in range(400):
            want = (want * 2 + want) % p
        _, ff, _ = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p, [], [2, 3])
        assert (ff is None) == ((want + p - 5) % p == 0)
        _, ff, _ = program_sim.simulate(ops, launches, consts, info['words_per_const'], info['slots'], p, [], [0, 5])
        assert ff is None                      # x = 0: acc stays 5, 5 + (p - 5) = 0
        info['working_entries'] = int(((ops[:, 1] & 0xFF) != 0).sum())    # (without the no-ops that keep a chain on one wave)
        results[drops] = info
    assert results[True]['slots'] < 20 < results[False]['slots']
    assert results[True]['working_entries'] < results[False]['working_entries']


# ---------------------------------------------------------------- GPU tier
@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(25, 45))
def test_streamed_random_relations_on_gpu(seed):
    """the fuzz of test_gpu_parity through a streamed ingest (windows of 32 calls, program uploaded window by window):
    violation strings, counts and the surviving top-level wires against the oracle"""
    p, boolean = FIELDS[seed % len(FIELDS)]
    g = Gen(seed, p, boolean)
    rel, mod_le = g.relation(n_top=14)
    lanes = 70
    rows_i, rows_w = g.lane_inputs(lanes, seed + 1000)
    ev = _streamed([rel], g.n_inst, g.n_wit, 32)
    ev.finalize()
    inst, wit = batch_arrays(rows_i, rows_w, ev.elem_bytes)
    ev.set_inputs(inst if g.n_inst else None, wit if g.n_wit else None, lanes)
    ev.replay()
    ev.synchronize()
    n_ok = 0
    for lane in range(0, lanes, 3):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32, trace=False)
        assert ev.get_violations(lane) == ref.violations, (seed, lane)
    first, _ = ev.lane_results(lanes)
    n_ok = int((first == zk.NO_FAIL).sum())
    assert ev.counts() == (n_ok, lanes - n_ok)
    ref = oracle_lane(mod_le, rows_i[0], rows_w[0], [rel], 32, trace=False)
    if not ref.violations:
        for wid in range(0, 60):
            want = ref.get(wid)
            got = ev.get(wid, lanes)
            assert (want is None) == (got is None), (seed, wid)
            if want is not None:
                assert got[0] == want, (seed, wid)


@pytest.mark.gpu
def test_full_size_c2_streamed_matches_the_committed_oracle_digests():
    """BASELINE configs[1] ingested as a stream (11 messages, windows of 131072 recorded calls scheduled and uploaded
    while the next messages are parsed): the 64 output wires of the six digest lanes are the oracle's, bit for bit."""
    import hashlib
    import json
    import os
    from helpers import ROOT
    fx = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'c2_digests.json')))
    wl = workloads.ArithLayered()
    batch = 1024
    inst, wit = wl.inputs(batch)
    ev = _streamed(wl.relation_messages(with_epilogue=False, free_last=False), wl.n_instance0, wl.n_witness, 1)
    ev.finalize()
    sinfo = ev.stream_info()
    assert sinfo['windows'] == 9 and sinfo['streamed_windows'] == 9
    ev.set_inputs(np.ascontiguousarray(inst[:, :wl.n_instance0]).tobytes(), wit.tobytes(), batch)
    ev.replay()
    ev.synchronize()
    cols = [ev.get(w, batch) for w in wl.output_wire_ids()]
    for lane, want in fx['lanes'].items():
        vals = [col[int(lane)] for col in cols]
        assert hashlib.sha256('\n'.join(str(v) for v in vals).encode()).hexdigest() == want['sha256'], lane


def test_scheduling_options_are_refused_once_a_streamed_schedule_has_started():
    """the streamed scheduler takes its options at the first window cut: a later zkgpu_set_option would be ignored by the
    windows already scheduled, so it is an error instead"""
    gates = [('witness', 0)] + [('addc', k, k - 1, bytes([1])) for k in range(1, 60)] + [('free', 0, 59)]
    rel = sw.write_relation(sw.int_to_le(101), 'arithmetic', 'simple', [], gates)
    ev = zk.Evaluator()
    ev.set_option('stream', '16')
    ev.set_option('fuse', '0')               # before the first Relation message: fine
    ev.declare_inputs(0, 1)
    ev.ingest_message(rel)
    for key in ('fuse', 'pair', 'fermat', 'propagate_copies', 'sort_by_operand', 'strand_width'):
        with pytest.raises(zk.ZkGpuError, match='the streamed schedule has started'):
            ev.set_option(key, '1')
    ev.set_option('streams', '1')            # a replay option: still fine
    ev.finalize()
    assert ev.stream_info()['windows'] > 1


def test_layered_relations_are_cut_at_their_level_seams():
    """A relation recorded level by level (the C2 / C4 shape) is cut in front of the first gate of a layer: no window ends
    inside a dependency level, so the streamed program has as many launches as the one scheduled at finalize (over GF(p) a
    gate whose reader lies behind a cut is not fused into it, which may cost a level of sources and a few slots; over GF(2)
    levels, launches and slots are the same)."""
    for wl in (workloads.ArithLayered(W=512, D=40, n_instance0=16, n_out=8), workloads.BoolLayered(W=512, D=40, n_instance0=16, n_out=8)):
        msgs = wl.relation_messages()
        whole = zk.Evaluator()
        whole.declare_inputs(wl.n_instance, wl.n_witness)
        for m in msgs:
            whole.ingest_message(m)
        whole.finalize()
        ev = _streamed(msgs, wl.n_instance, wl.n_witness, 2000)
        ev.finalize()
        assert ev.stream_info()['windows'] >= 8 and ev.stream_info()['streamed_windows'] == ev.stream_info()['windows']
        a, b = whole.schedule_info(), ev.schedule_info()
        assert a['launches'] == b['launches'], (wl.p, a, b)
        if wl.p == 2:
            assert (a['levels'], a['slots'], a['device_ops']) == (b['levels'], b['slots'], b['device_ops']), (a, b)


@pytest.mark.parametrize('split', ['messages', 'one_buffer'])
def test_streamed_gf2_relation_against_the_oracle_and_the_unstreamed_lds_program(split):
    """GF(2) relations stream like the others (evaluator.rs:286-301 consumes every relation message by message): the
    windows are scheduled while the messages come in, the program of the LDS-resident kernel is built from the finished
    schedule -- for a relation recorded level by level it is the program of the schedule made at finalize, byte for byte,
    however the relation was split into messages -- and its verdicts are the oracle's."""
    from test_lds_program import interpret
    wl = workloads.BoolLayered(W=2048, D=12, n_instance0=64, n_out=16)
    inst, wit = wl.inputs(8)
    probe = cpu_checkers_outputs(wl, inst, wit)
    wl.set_expected_outputs(inst, probe, corrupt_every=3)
    msgs = wl.relation_messages()
    parts = msgs if split == 'messages' else [b''.join(msgs)]
    ev = _streamed(parts, wl.n_instance, wl.n_witness, 4096)
    ev.finalize()
    assert ev.stream_info()['windows'] >= 4 and ev.stream_info()['streamed_windows'] == ev.stream_info()['windows']
    whole = zk.Evaluator()
    whole.declare_inputs(wl.n_instance, wl.n_witness)
    for m in msgs:
        whole.ingest_message(m)
    whole.finalize()
    P, Q = ev.lds_program(0), whole.lds_program(0)
    for k in ('ops8', 'rows', 'blocks', 'chunks'):
        assert np.array_equal(P[k], Q[k]), k
    assert P['block_rows'] == Q['block_rows'] and P['table_words'] == Q['table_words']
    first = interpret(P, ev.schedule_dump()[2], inst[:, :, 0], wit[:, :, 0])
    for lane in range(8):
        ref = oracle_lane(wl.mod_le, [int(x) for x in inst[lane, :, 0]], [int(x) for x in wit[lane, :, 0]], msgs, 1, trace=False)
        assert (first[lane] == 0xFFFFFFFF) == (ref.violations == []) == (lane % 3 != 0), lane


def cpu_checkers_outputs(wl, inst, wit):
    import cpu_checkers
    return cpu_checkers.bool_layered_outputs(wl, inst, wit)


def test_a_byte_stream_of_several_messages_is_decoded_ahead_with_the_same_outcome():
    """zkgpu_ingest_messages on a buffer that holds several messages decodes message k + 1 on a helper thread while message
    k is recorded (capi.cpp MessageDecoder).  Outcome by outcome what ingesting them one call at a time gives: the same
    tape, the same latch when a message in the middle does not decode (the reference panics there, evaluator.rs:193), and
    nothing recorded behind it."""
    wl = workloads.ArithLayered(W=256, D=60, n_instance0=16, n_out=8)
    msgs = wl.relation_messages()
    msgs = msgs + workloads.ArithLayered(W=64, D=2, n_instance0=4, n_out=2).relation_messages()[:0]
    parts = [sw.write_relation(wl.mod_le, 'arithmetic', 'simple', [], [('witness', 1000 + k)] + [('mul', 2000 + 3 * k + j, 1000 + k, 1000 + k) for j in range(3)])
             for k in range(6)]

    def session(bufs):
        ev = zk.Evaluator()
        ev.declare_inputs(wl.n_instance, wl.n_witness + 6)
        for b in bufs:
            ev.ingest_message(b)
        return ev
    one_by_one, at_once = session(msgs + parts), session([b''.join(msgs + parts)])
    assert at_once.host_violations() == one_by_one.host_violations() == []
    for x, y in zip(at_once.tape(), one_by_one.tape()):
        assert np.array_equal(x, y)
    # a message that does not decode, in the middle: everything before it is recorded, the error latches, nothing behind it
    broken = bytearray(parts[2])
    broken[40:48] = b'\xff' * 8
    bad = msgs + parts[:2] + [bytes(broken)] + parts[3:]
    a, b = session(bad), session([b''.join(bad)])
    assert a.host_violations() == b.host_violations() and a.host_violations() != []
    assert a.tape_len == b.tape_len == session(msgs + parts[:2]).tape_len
