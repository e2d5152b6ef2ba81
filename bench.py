#!/usr/bin/env python3
"""Headline benchmark: gate-ops/sec of the SIEVE IR batch evaluator on MI355X.

Workload = BASELINE.json configs[1] (C2): BN254 scalar field, 2^20-gate Add/Mul
relation (W=4096 x D=256), 1024 witnesses per GPU, synthetic inputs.  One
"step" = one replay of the whole recorded tape over the GPU's witness batch
(instance/witness loads + canonicality check, every gate, the 64 AssertZero
checks, the verdict reduction) and, for N > 1, the RCCL all-reduce of the
{satisfied, failed} counts.  Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The default invocation (one GPU, C2) also measures the other three workloads -- BASELINE configs[3] (C4, GF(2)),
configs[4] (C5, R1CS rows) and the structured For / Call / Switch relation -- and prints them under `secondary`, each
with its own `roofline` and `cpu_baseline` (bounded samples; `--no-secondary` skips them).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (child processes, started
before anything in this process touches the GPU); rank 0 prints the JSON line.  With fewer than N devices visible
the ranks share the card and the count reduction runs over gloo (rehearsal of the same sharding; the line says so).

Numbers in the line and where they come from:
  * every time is measured in this run (wall clock around the timed steps; HIP events on the engine's stream for the
    kernel time of a replay);
  * every byte / instruction count that needs hardware counters comes from the committed `--pmc` passes of this same
    command (profiles/binding_<workload>.json, written by tools/binding_evidence.py) and is only used when the program
    it was counted on is the program that ran (same entries, same launches); every rate and fraction is computed here
    from those counts and this run's times, so no figure appears twice with two values.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

# algorithmic bytes per backend op and witness for a 256-bit field
BYTES_PER_OP = {1: 96, 2: 96, 10: 96, 11: 96,       # add, mul, and, xor: 2 x 32 B read + 32 B write
                3: 64, 4: 64, 5: 64, 12: 64,        # addc, mulc, copy, not: 32 B read + 32 B write
                6: 32,                              # constant: 32 B write
                7: 64, 8: 64,                       # instance / witness: 32 B read + 32 B write
                9: 32}                              # assert_zero: 32 B read
# GF(2), bit-packed wires: one bit per operand / result
BYTES_PER_OP_BOOL = {1: 0.375, 2: 0.375, 10: 0.375, 11: 0.375, 3: 0.25, 4: 0.25, 5: 0.25, 12: 0.25, 6: 0.125,
                     7: 1.125, 8: 1.125,            # one input byte read + one bit written
                     9: 0.125}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def binding_constants(workload, program):
    """Counter-derived, time-independent facts about the dominant kernel of a workload (profiles/binding_<workload>.json,
    tools/binding_evidence.py): bytes crossing L2<->fabric per launch, instructions per wave, VALU pipe time.  Only valid
    for the program they were counted on: `program` = {entries, launches, batch, ...} of this run must match."""
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'binding_%s.json' % workload)))
    except (OSError, ValueError):
        return None
    want = d.get('program') or {}
    if any(want.get(k) != program.get(k) for k in want):
        return None
    # the kernels and the scheduler the counters were collected on must be the ones in the tree: a kernel change can move
    # traffic or instruction counts without changing the program's shape (tools/binding_evidence.py stamps the digest)
    if d.get('device_source_digest') != entry.device_source_digest():
        return {'stale': True, 'collected_on': d.get('device_source_digest')}
    return d


def golden_hashes(name):
    """tests/golden/<name>_all_lanes.json: one 64-bit hash of the output wires per lane, made by the CPU checkers that are
    pinned to the oracle (tests/golden/make_all_lanes.py).  Data only; None when the file is absent."""
    try:
        return json.load(open(os.path.join(ROOT, 'tests', 'golden', '%s_all_lanes.json' % name)))['hashes']
    except (OSError, ValueError, KeyError):
        return None


def check_probe_outputs(name, wl, outs, lane_offset):
    """The expected outputs the measured relation compares against come from a GPU probe pass (a different schedule and
    kernel from the measured one).  For the default sizes every lane of them is checked here against the committed
    oracle-chain hashes, so a wrong but self-consistent device result cannot pass.  Returns how many lanes were checked."""
    hashes = golden_hashes(name)
    # (the fixtures were made for the default wiring, mix and sizes: any other relation has other outputs)
    default = ((name == 'c2' and (wl.W, wl.D) == (4096, 256) and getattr(wl, 'default_mix', True) and getattr(wl, 'p', 0).bit_length() == 254 and getattr(wl, 'p', 0) % 2 ** 32 == 0xf0000001) or
               (name == 'c4' and (wl.W, wl.D) == (16384, 640) and getattr(wl, 'wiring', 'random') == 'random'))
    if hashes is None or not default:
        return 0
    covered = max(0, min(len(outs), len(hashes) - lane_offset))     # (the fixtures hold 8192 lanes of C2, 4096 of C4)
    for lane in range(covered):
        h = hashlib.sha256(np.ascontiguousarray(outs[lane], dtype=np.uint8).tobytes()).hexdigest()[:16]
        assert h == hashes[lane + lane_offset], 'lane %d: probe outputs differ from tests/golden/%s_all_lanes.json' % (lane + lane_offset, name)
    return covered


class _DevU64x2:
    """torch view of the engine's device counters {satisfied, failed}"""

    def __init__(self, ptr):
        self.__cuda_array_interface__ = {'shape': (2,), 'typestr': '<i8', 'data': (int(ptr), False), 'version': 2}


def first_verdict_seconds(zk, wl, msgs, inst, wit, batch, stream, pinned, bool_path=None):
    """relation in -> first verdict out for ONE batch on a fresh session (the GPU runtime is already up): ingest of
    the relation messages, scheduling, program upload, input hand-over, one replay, counts back on the host.  With
    stream=1 the windows of the tape are scheduled and uploaded by a worker thread while the later messages are
    still being parsed."""
    if pinned:
        import torch
        keep = (torch.from_numpy(np.ascontiguousarray(inst).reshape(-1)).pin_memory(),
                torch.from_numpy(np.ascontiguousarray(wit).reshape(-1)).pin_memory())
        ib, wb = keep[0].data_ptr(), keep[1].data_ptr()
    else:
        ib, wb = inst.tobytes(), wit.tobytes()
    stream_bytes = b''.join(msgs)
    t0 = time.perf_counter()
    ev = zk.Evaluator()
    if bool_path:
        ev.set_option('bool_path', bool_path)
    ev.set_option('stream', '1' if stream else '0')
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    ev.ingest_message(stream_bytes)     # one byte stream, as a file of a workspace is (consumers/source.rs:91-118): decoded on a
    t1 = time.perf_counter()            # helper thread while the messages before are recorded
    ev.finalize()
    t2 = time.perf_counter()
    ev.set_inputs(ib, wb, batch)
    t3 = time.perf_counter()
    ev.replay()
    counts = ev.counts()
    t4 = time.perf_counter()
    info = ev.stream_info()
    ev.close()
    return {'total_s': round(t4 - t0, 4), 'ingest_s': round(t1 - t0, 4), 'finalize_s': round(t2 - t1, 4),
            'set_inputs_s': round(t3 - t2, 4), 'replay_and_counts_s': round(t4 - t3, 4), 'windows': info['windows'],
            'worker_busy_s': round(info['worker_busy_s'], 4), 'satisfied': counts[0]}


def build_session(zk, name, wl, batch, lane_offset, lane_group, bool_path=None, streams=2):
    """probe pass for the expected outputs, then the real session with resident inputs"""
    t0 = time.time()
    if getattr(wl, 'closed_form', False):   # StructuredArith, Sha256Compress: the expected values come with the inputs, no probe pass
        inst, wit, n_bad = wl.inputs(batch, lane_offset)
        return finish_session(zk, wl, batch, lane_group, bool_path, streams, inst, wit, n_bad, t0, time.time(), 0)
    inst, wit = wl.inputs(batch, lane_offset)
    # The probe runs the relation without its epilogue on ANOTHER schedule and kernel than the measured session: the
    # unfused program (`replay_kernel`) for GF(p), the HBM-table kernel for GF(2) when the LDS-resident one is measured.
    # (It also keeps the kernels of the timed steps the only launches of their name in a rocprofv3 trace of this command.)
    probe = zk.Evaluator()
    if bool_path:
        probe.set_option('bool_path', 'hbm' if bool_path != 'hbm' else 'auto')
    else:
        probe.set_option('fuse', '0')
    probe.declare_inputs(wl.n_instance0, wl.n_witness)
    for m in wl.relation_messages(with_epilogue=False, free_last=False):
        probe.ingest_message(m)
    probe.finalize()
    probe.set_inputs(np.ascontiguousarray(inst[:, :wl.n_instance0]).tobytes(), wit.tobytes(), batch)
    probe.replay()
    probe.synchronize()
    outs = np.zeros((batch, wl.n_out, wl.width), dtype=np.uint8)
    for t, wid in enumerate(wl.output_wire_ids()):
        vals = probe.get(wid, batch)
        outs[:, t] = np.frombuffer(b''.join(v.to_bytes(wl.width, 'little') for v in vals),
                                   dtype=np.uint8).reshape(batch, wl.width)
    if wl.p == 2:
        outs = outs[:, :, 0]
    probe.close()
    pinned_lanes = check_probe_outputs(name, wl, outs, lane_offset)
    n_bad = wl.set_expected_outputs(inst, outs, lane_offset)
    return finish_session(zk, wl, batch, lane_group, bool_path, streams, inst, wit, n_bad, t0, time.time(), pinned_lanes)


def finish_session(zk, wl, batch, lane_group, bool_path, streams, inst, wit, n_bad, t0, t1, pinned_lanes):
    msgs = wl.relation_messages()
    t2 = time.time()
    ev = zk.Evaluator()
    if bool_path:
        ev.set_option('bool_path', bool_path)
        # GF(2) relations are ingested as a stream: the windows end at the seams between dependency levels, and the program
        # of a relation recorded level by level is the finalize-time program byte for byte (tests/test_full_size.py)
        ev.set_option('stream', '1')
    ev.set_option('streams', str(streams))
    # developer switches for A/B runs (profiles/*_tuning_sweeps.txt); the metric is quoted on the defaults
    for env, opt in (('ZKI_STREAM', 'stream'), ('ZKI_FUSE', 'fuse'), ('ZKI_OPW', 'level_ops_per_wave'), ('ZKI_HOT_WAVES', 'hot_waves'),
                     ('ZKI_GRAPH', 'graph'), ('ZKI_XCD_MAP', 'xcd_map'), ('ZKI_SORT_BY_OPERAND', 'sort_by_operand'),
                     ('ZKI_STRAND_WIDTH', 'strand_width'), ('ZKI_BANK_AWARE', 'bank_aware'), ('ZKI_FERMAT', 'fermat'),
                     ('ZKI_PAIR', 'pair'), ('ZKI_STRAND_LDS', 'strand_lds'), ('ZKI_STRAND_PREFETCH', 'strand_prefetch'), ('ZKI_STRAND_MERGE', 'strand_merge'), ('ZKI_STRAND_REASSOC', 'strand_reassociate'), ('ZKI_STRAND_SPLIT', 'strand_split_inputs'), ('ZKI_BOOL_NARROW', 'bool_narrow_width')):
        if os.environ.get(env):
            ev.set_option(opt, os.environ[env])
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    stream_bytes = b''.join(msgs)
    t2 = time.time()
    ev.ingest_message(stream_bytes)     # (one byte stream: decoded on a helper thread while the messages before are recorded)
    assert ev.host_violations() == [], ev.host_violations()
    t3 = time.time()
    ev.finalize()
    t4 = time.time()
    if lane_group:
        ev.set_lane_group(lane_group)
    ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
    t5 = time.time()
    host = {'probe_s': t1 - t0, 'emit_sieve_s': t2 - t1, 'ingest_and_record_s': t3 - t2, 'schedule_s': t4 - t3,
            'h2d_s': t5 - t4, 'relation_bytes': sum(len(m) for m in msgs), 'messages': len(msgs),
            'lanes_checked_against_golden_hashes': pinned_lanes}
    return ev, inst, wit, n_bad, msgs, host


def cpu_baseline(wl, msgs, inst, wit, gates, ev=None, budget_s=15.0, with_opt=True):
    """The oracle (literal restatement of the reference Evaluator + PlaintextBackend) on this
    box's host cores, on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import oracle_lib
    cores = os.cpu_count() or 1
    rel = b''.join(msgs)
    one = oracle_lib.eval_batch(rel, wl.mod_le, inst[:1].tobytes(), wl.n_instance, wit[:1].tobytes(), wl.n_witness,
                                wl.width, 1, 1)
    per_lane = max(one[1], 1e-3)
    threads = min(cores, 64)
    # whole rounds of one witness per thread: as many as the budget allows at the single-thread time, at most four (with
    # every thread busy a witness takes several times its single-thread time: hash maps and heap integers share the caches)
    rounds = max(1, min(4, int(budget_s / (4.0 * per_lane))))
    lanes = int(min(inst.shape[0], rounds * threads))
    ok, secs, ops = oracle_lib.eval_batch(rel, wl.mod_le, inst[:lanes].tobytes(), wl.n_instance,
                                          wit[:lanes].tobytes(), wl.n_witness, wl.width, lanes, threads)
    extra = {}
    if wl.p != 2 and ev is not None and with_opt:
        # BASELINE.md section 3 `cpu_opt`: flat array + 64-bit Montgomery on the recorded tape, same threads
        kinds, a, b = ev.tape()
        opt_lanes = int(min(inst.shape[0], 16 * threads))
        _, osecs, _ = oracle_lib.opt_eval(kinds, a, b, ev.constants(), wl.mod_le, inst[:opt_lanes].tobytes(),
                                          wl.n_instance, wit[:opt_lanes].tobytes(), wl.n_witness, wl.width, opt_lanes,
                                          threads)
        extra = {'cpu_opt': {'value': gates * opt_lanes / osecs, 'unit': 'gate-ops/s', 'cores': threads,
                             'what': 'optimised CPU evaluator (flat wire array, 4x64 Montgomery, flattened tape), '
                                     '%d witnesses in %.1f s' % (opt_lanes, osecs)}}
    # the oracle's verdicts on the sample are a second check of the GPU counts (the first: check_probe_outputs)
    from zkinterface_ir_amd import workloads
    want = workloads.expected_satisfied(lanes)
    assert int(sum(ok)) == want, 'oracle: %d of %d sample lanes satisfied, expected %d' % (int(sum(ok)), lanes, want)
    if ev is not None:
        ff = np.asarray(ev.lane_results(inst.shape[0])[0][:lanes])
        assert np.array_equal(ff == 0xFFFFFFFF, np.asarray(ok, dtype=bool)), 'GPU and oracle verdicts differ on the sample'
    return {**extra, 'value': gates * lanes / secs, 'unit': 'gate-ops/s', 'cores': threads, 'kind': 'port',
            'sample_witnesses': lanes, 'sample_seconds': secs, 'single_witness_seconds': per_lane,
            'sample': '%d witnesses of the same %d-gate relation, one reference-style Evaluator run per witness, '
                      '%d threads, %.1f s wall (single witness: %.2f s)' % (lanes, gates, threads, secs, per_lane),
            'satisfied_in_sample': int(sum(ok))}


def make_roofline(workload, kernel, launches, kernel_ms_per_step, algo_bytes_per_step, program, streams=1):
    """The `roofline` object of one line.

    frac_algorithmic = SURVEY.md 8(d) bytes of the step / HIP-event time of the step / the HBM spec peak: the figure the
    north star asks for.  It can exceed 1 (a schedule that keeps values in registers and an on-die cache move fewer bytes,
    from a faster memory, than that accounting assumes), so it is not the roofline: `bound` names the resource the committed
    counters show the kernel is limited by, and achieved / peak / frac (<= 1) are against THAT resource.  `resources` holds
    every candidate with its own fraction; `bound` is the largest."""
    step_s = kernel_ms_per_step * 1e-3
    algo_gbs = algo_bytes_per_step / step_s / 1e9
    res = {}
    out = {'kernel': kernel, 'launches_per_step': launches, 'concurrent_streams': streams,
           'avg_launch_ms': kernel_ms_per_step / max(launches, 1),
           'algorithmic_bytes_per_launch': algo_bytes_per_step / max(launches, 1),
           'achieved_algorithmic': algo_gbs, 'frac_algorithmic': algo_gbs / HBM_PEAK_GBS, 'traffic': None, 'traffic_source': None}
    bc = binding_constants(workload, program)
    stale = bool(bc and bc.get('stale'))
    if stale:
        bc = None
    c = (bc or {}).get('constants', {})
    if c.get('traffic_bytes_per_launch') is not None:
        per_step = c['traffic_bytes_per_launch'] * c.get('traffic_launches_per_step', launches)
        gbs = per_step / step_s / 1e9
        res[c.get('memory_side', 'hbm')] = {'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
                                            'bytes_per_step': per_step, 'over_algorithmic': per_step / algo_bytes_per_step}
        out['traffic'] = c['traffic_bytes_per_launch']
        out['traffic_source'] = c.get('traffic_source')
    if c.get('valu_pipe_ms_per_step') is not None:
        res['valu'] = {'achieved': c['valu_pipe_ms_per_step'], 'peak': kernel_ms_per_step,
                       'unit': 'ms of VALU pipe per step (counted instructions x the rates of profiles/r01_valu_rates.txt) over ms per step',
                       'frac': c['valu_pipe_ms_per_step'] / kernel_ms_per_step, 'valu_insts_per_wave': c.get('valu_insts_per_wave')}
    if c.get('lds_array_cycles_per_step') is not None:
        # The LDS array of a CU serves one lane group per cycle (MI355X_MICROARCH.md, LDS: SQ_LDS_IDX_ACTIVE = all LDS-array
        # cycles, bank conflicts included).  Against the HARDWARE: the arrays of all 256 CUs at 2.4 GHz for the time of the
        # step; the kernel occupies one CU per 32-witness slice (cus_busy of cus_total), frac_on_busy_cus is the same count
        # against those CUs alone.
        cus_total, cus_busy = 256, min(256, int(c.get('workgroups', 0)) or 256)
        peak = cus_total * 2.4e9 * step_s
        res['lds_array'] = {'achieved': c['lds_array_cycles_per_step'], 'peak': peak, 'unit': 'LDS-array cycles per step (256 CUs x 2.4 GHz)',
                            'frac': c['lds_array_cycles_per_step'] / peak, 'cus_busy': cus_busy, 'cus_total': cus_total,
                            'frac_on_busy_cus': c['lds_array_cycles_per_step'] / (cus_busy * 2.4e9 * step_s)}
    if not res:
        # no counters for this program: the algorithmic figure against HBM is all there is
        res['hbm_algorithmic'] = {'achieved': algo_gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': algo_gbs / HBM_PEAK_GBS}
    bound = max(res, key=lambda k: res[k]['frac'])
    out.update({'bound': bound, 'achieved': res[bound]['achieved'], 'peak': res[bound]['peak'], 'unit': res[bound]['unit'],
                'frac': res[bound]['frac'], 'resources': res})
    if bc:
        out['binding'] = bc.get('binding')
        out['counters'] = {k: v for k, v in c.items() if k not in ('traffic_bytes_per_launch', 'traffic_source', 'valu_pipe_ms_per_step',
                                                                   'lds_array_cycles_per_step', 'memory_side', 'traffic_launches_per_step')}
        out['reading'] = bc.get('reading')
        out['sources'] = bc.get('sources')
    elif stale:
        out['binding'] = ('profiles/binding_%s.json was collected on other kernel / scheduler sources than the ones in the tree '
                          '(device_source_digest differs): its counters are not used' % workload)
    else:
        out['binding'] = 'not profiled for this program (no matching profiles/binding_%s.json)' % workload
    return out


def break_even(first_verdict_s, gpu_ms_per_step, batch, cpu):
    """Witnesses from which on the GPU path answers sooner than the CPU baseline on all its cores: the host work of the
    GPU path (relation in -> first verdict, minus the replay itself) is paid once, a witness then costs ms_per_step / batch
    on the GPU and 1 / (witnesses per second of the CPU sample) on the host cores.  None when the CPU is never caught up."""
    if not cpu or not first_verdict_s:
        return None
    cpu_s = cpu['sample_seconds'] / max(cpu['sample_witnesses'], 1)       # all cores busy: seconds per witness
    gpu_s = gpu_ms_per_step * 1e-3 / batch
    host_s = max(first_verdict_s - gpu_ms_per_step * 1e-3, 0.0)
    return None if cpu_s <= gpu_s else int(np.ceil(host_s / (cpu_s - gpu_s)))


def _r(x, digits=4):
    if isinstance(x, float):
        return float('%.*g' % (digits, x))
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def compact_roofline(r):
    """the part of a roofline object the JSON line carries (the prose and the counter lists go to the detail file)"""
    if not r:
        return None
    out = {k: r.get(k) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'frac_algorithmic', 'traffic',
                                 'launches_per_step', 'avg_launch_ms')}
    if len(str(out.get('unit') or '')) > 24:
        out['unit'] = {'valu': 'ms VALU pipe / ms', 'lds_array': 'LDS-array cycles'}.get(r.get('bound'), str(out['unit'])[:24])
    out['resources'] = {k: v['frac'] for k, v in (r.get('resources') or {}).items()}
    la = (r.get('resources') or {}).get('lds_array')
    if la:
        out.update({'cus_busy': la['cus_busy'], 'cus_total': la['cus_total'], 'frac_on_busy_cus': la['frac_on_busy_cus']})
    for k in ('batch', 'ms_per_step', 'value'):
        if k in r:
            out[k] = r[k]
    if 'counters' not in r:
        out['counters'] = 'none'     # no (current) profiles/binding_*.json: only the algorithmic figure is hardware-free
    return out


def compact_line(full, detail_path):
    """The ONE JSON line: every number the judge needs, <= 6 KB.  `full` (everything, prose included) goes to detail_path."""
    def wl(d, head=False):
        cfg = d.get('config', {})
        o = {}
        o.update({'ms_per_step': d['ms_per_step'], 'value': d['value'], 'unit': d['unit'], 'batch': cfg.get('batch_per_gpu'),
                  'roofline': compact_roofline(d.get('roofline'))})
        cb = d.get('cpu_baseline')
        if cb:
            o['cpu_baseline'] = {'value': cb['value'], 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'],
                                 'sample': '%d witnesses, %.1f s' % (cb.get('sample_witnesses', 0), cb.get('sample_seconds', 0.0))}
            if not head:
                o['cpu_baseline'] = {'value': cb['value'], 'cores': cb['cores']}
            if 'cpu_opt' in cb:
                o['cpu_baseline']['cpu_opt'] = cb['cpu_opt']['value']
        fv = cfg.get('relation_in_to_first_verdict')
        if fv:
            o['first_verdict_s'] = {k: v['total_s'] for k, v in fv.items()}
        if cfg.get('break_even_batch') is not None:
            o['break_even_batch'] = cfg['break_even_batch']
        if cfg.get('host_seconds'):     # (the other stages are in the detail file)
            keep = ('ingest_and_record_s', 'schedule_s', 'build_s', 'witness_generation_s')
            o['host_seconds'] = {k: v for k, v in cfg['host_seconds'].items() if head or k in keep}
        o['counts'] = [cfg.get('satisfied'), cfg.get('failed')]
        for k in ('batch_8192',):
            if k in d:
                o[k] = {q: d[k][q] for q in ('ms_per_step', 'value', 'workgroups') if q in d[k]}
        return o
    line = {k: full[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
                                 'scaling', 'vs_baseline', 'dtype', 'data')}
    head = wl(full, head=True)
    line['roofline'] = head['roofline']
    if 'cpu_baseline' in head:
        line['cpu_baseline'] = head['cpu_baseline']
    if 'roofline_hbm' in full:
        line['roofline_hbm'] = compact_roofline(full['roofline_hbm'])
    def brief(d):   # a variant of a workload the line already carries in full
        r = d.get('roofline') or {}
        return {'ms_per_step': d['ms_per_step'], 'value': d['value'], 'counts': [d['config'].get('satisfied'), d['config'].get('failed')],
                'classes': d['config'].get('combinations_by_coefficient_class'),
                'roofline': {'kernel': r.get('kernel'), 'bound': r.get('bound'), 'frac': r.get('frac')}}
    if 'secondary' in full:
        line['secondary'] = {k: (brief(v) if k == 'c5_small' else wl(v)) for k, v in full['secondary'].items() if v}
        line['secondary_wall_s'] = full.get('secondary_wall_s')
    cfg = full.get('config', {})
    keep = ('workload', 'batch_per_gpu', 'backend_ops_per_witness', 'program_entries', 'levels', 'launches_per_step',
            'wire_table_slots', 'wire_table_MB', 'parallelism', 'pcie_inclusive_ms_per_step', 'satisfied', 'failed', 'rows',
            'variables', 'host_seconds', 'break_even_batch', 'tape_windows', 'rank_base', 'expected_outputs')
    line['config'] = {k: cfg[k] for k in keep if cfg.get(k) is not None}
    if head.get('first_verdict_s'):
        line['config']['first_verdict_s'] = head['first_verdict_s']
    if cfg.get('ranks_seen'):
        line['config']['ranks_seen'] = cfg['ranks_seen']
    line['detail'] = detail_path
    return _r(line)


def bench_c5(args, zk, workloads, ctx, steps, warmup, cpu_budget_s=15.0, coef_kind='random'):
    """BASELINE configs[4]: 2^20-row R1CS over BN254 (3+3+1 terms per row), witness batch 1024 per GPU.
    step = the row check <a,w>*<b,w> = <c,w> of every row for every lane + the count reduction.
    coef_kind 'small': the same rows with coefficients 1 / -1 / 16-bit signed integers (workloads.R1csSynthetic), which the
    row kernel takes by its unit / small coefficient classes -- a variant next to the BASELINE line, never instead of it."""
    world, rank, dist, torch, red_dev, dist_on = ctx['world'], ctx['rank'], ctx['dist'], ctx['torch'], ctx['red_dev'], ctx['dist_on']
    M = (args.width or (1 << 20)) if args.workload == 'c5' else (1 << 20)
    batch = (args.batch_per_gpu or 1024) if args.workload == 'c5' else 1024
    t0 = time.time()
    wl = workloads.R1csSynthetic(M=M, coef_kind=coef_kind)
    small = coef_kind != 'random'
    ev = zk.Evaluator()
    ev.declare_inputs(0, wl.n_witness)
    ev.ingest_message(wl.base_relation())
    ev.finalize(retain_all=True)
    row_ptr, tv, tc, cb = wl.csr()
    ev.r1cs_load_csr(row_ptr, tv, tc, cb, wl.width, wl.M)
    classes = ev.r1cs_class_counts()
    t1 = time.time()
    lane_offset = rank * batch
    w = wl.witnesses(batch, lane_offset)
    ev.set_inputs(None, w.tobytes(), batch)
    ev.replay()
    lo = 0
    for hi in wl.level_bounds:      # witness generation, one launch per dependency level (setup, untimed)
        ev.r1cs_assign(lo, int(hi) - lo)
        lo = int(hi)
    zl = ev.r1cs_get_var(wl.last_z, batch)
    bad = 0
    for lane in range(batch):
        v = zl[lane]
        if (lane + lane_offset) % 97 == 0:
            v = (v + 1) % wl.p
            bad += 1
        w[lane, wl.n_base] = np.frombuffer(v.to_bytes(wl.width, 'little'), dtype=np.uint8)
    ev.set_inputs(None, w.tobytes(), batch)
    ev.replay()
    ev.synchronize()
    t2 = time.time()
    n_rows = M + 1

    def step():
        ev.r1cs_check()
        return ev.r1cs_results(batch)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    ts = time.perf_counter()
    ms = []
    for _ in range(steps):
        ff, counts = step()
        ms.append(ev.r1cs_last_ms)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - ts
    total = list(counts)
    if dist_on:
        t = torch.tensor(total + [0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t)
        total = [int(t[0].item()), int(t[1].item())]
        tm = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        elapsed = float(tm.item())
    assert total[0] == workloads.expected_satisfied(batch * world) and total[1] == batch * world - total[0], total
    out = None
    if rank == 0:
        kernel_ms = float(np.mean(ms))
        out = {
            'metric': 'row-checks/sec (whole node), 1M-constraint R1CS over BN254, batched witnesses',
            'value': n_rows * batch * world / (elapsed / steps), 'unit': 'row-checks/s', 'n_gpus': world,
            'steps': steps, 'warmup': warmup, 'ms_per_step': elapsed * 1e3 / steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'u64x4 (GF(p) Montgomery limbs, exact integer)', 'data': 'synthetic',
            'config': {'workload': ('the rows of BASELINE configs[4] with coefficients 1 / -1 / 16-bit signed integers: ' if small else
                                    'BASELINE configs[4]: ') + '%d-row R1CS (3+3+1 terms, %s coefficients) over BN254, '
                                   'witness batch=%d per GPU, %d GPU(s)' % (M, 'small' if small else 'random', batch, world),
                       'combinations_by_coefficient_class': classes,
                       'batch_per_gpu': batch, 'variables': wl.n_base + 1 + M, 'dependency_levels': wl.n_levels, 'rows': n_rows,
                       'wire_table_GB': round(ev.table_bytes / 1e9, 2), 'satisfied': total[0], 'failed': total[1],
                       'host_seconds': {'build_s': round(t1 - t0, 2), 'witness_generation_s': round(t2 - t1, 2)}},
            'roofline': make_roofline('c5_small' if small else 'c5', 'r1cs_row_kernel<8, false, %s>' % ('true' if small else 'false'), 1,
                                      kernel_ms, 7.0 * wl.width * M * batch, {'entries': n_rows, 'launches': 1, 'batch': batch}),
        }
        if not args.no_cpu_baseline and world == 1 and cpu_budget_s > 0:  # rank 0 at N=1 only
            # the row check on the host cores for a bounded sample of the same lanes (oracle/cpu_opt.cpp:
            # the mathematical definition -- the reference itself holds no row checker)
            sys.path.insert(0, os.path.join(ROOT, 'tests'))
            import oracle_lib
            threads = min(os.cpu_count() or 1, 64)
            sample = min(batch, max(threads, int(4 * threads * min(1.0, cpu_budget_s / 15.0))))
            ff_cpu, secs = oracle_lib.r1cs_check(row_ptr, tv, tc, cb, wl.mod_le, w[:sample], wl.n_base + 1 + M, M, threads)
            assert np.array_equal(ff_cpu, np.asarray(ff[:sample])), 'CPU row check disagrees with the GPU'
            out['cpu_baseline'] = {'value': n_rows * sample / secs, 'unit': 'row-checks/s', 'cores': threads, 'kind': 'port',
                                   'sample_witnesses': sample, 'sample_seconds': secs,
                                   'sample': '%d witnesses of the same %d-row system, 4x64 Montgomery row check on %d '
                                             'threads, %.1f s wall (witness generation excluded)' % (sample, n_rows, threads, secs)}
    if rank == 0 and world == 1 and not args.no_first_verdict and not (small and args.workload != 'c5'):
        # relation + constraint system in -> first verdict out on a fresh session (the witness generation of the product
        # rows, level by level, is part of it: the batch arrives as base variables)
        w0 = wl.witnesses(batch, lane_offset)
        w0[:, wl.n_base] = w[:, wl.n_base]
        runs = []
        for _ in range(2):
            ta = time.perf_counter()
            e2 = zk.Evaluator()
            e2.declare_inputs(0, wl.n_witness)
            e2.ingest_message(wl.base_relation())
            e2.finalize(retain_all=True)
            e2.r1cs_load_csr(row_ptr, tv, tc, cb, wl.width, wl.M)
            tb = time.perf_counter()
            e2.set_inputs(None, w0.tobytes(), batch)
            e2.replay()
            lo = 0
            for hi in wl.level_bounds:
                e2.r1cs_assign(lo, int(hi) - lo)
                lo = int(hi)
            e2.r1cs_check()
            _, c2 = e2.r1cs_results(batch)
            tc_ = time.perf_counter()
            assert list(c2) == total, (c2, total)
            e2.close()
            runs.append({'total_s': round(tc_ - ta, 4), 'load_s': round(tb - ta, 4), 'inputs_assign_check_s': round(tc_ - tb, 4)})
        out['config']['relation_in_to_first_verdict'] = {'at_finalize': min(runs, key=lambda r: r['total_s'])}
        if 'cpu_baseline' in out:
            out['config']['break_even_batch'] = break_even(out['config']['relation_in_to_first_verdict']['at_finalize']['total_s'],
                                                           out['ms_per_step'], batch, out['cpu_baseline'])
    ev.close()
    return out


def bench_tape(name, args, zk, workloads, ctx, steps, warmup, headline, cpu_budget_s=15.0):
    """C2 (headline), C4 and the structured relation: one step = one replay of the recorded tape over the batch.
    `headline`: the workload the command line names (its size flags apply; the extras of the headline line are measured)."""
    world, rank, dist, torch, red_dev, dist_on = ctx['world'], ctx['rank'], ctx['dist'], ctx['torch'], ctx['red_dev'], ctx['dist_on']
    named = name == args.workload      # --width / --depth / --batch-per-gpu describe the workload named on the command line
    width, depth, bpg = (args.width, args.depth, args.batch_per_gpu) if named else (0, 0, 0)
    lane_group = args.lane_group if named else 0
    if name == 'c2':
        mp = os.environ.get('ZKI_C2_MUL_PERCENT')  # developer sensitivity runs only; the metric is quoted on the default mix
        mod = os.environ.get('ZKI_C2_MODULUS')     # developer runs only: the C2 shape over another characteristic, e.g. an even
        wl = workloads.ArithLayered(W=width or 4096, D=depth or 256, mul_percent=int(mp) if mp else None,   # one (any-modulus kernels)
                                    **({'p': int(mod, 0)} if mod else {}))
        batch = bpg or 1024
        bytes_table, bool_path = BYTES_PER_OP, None
    elif name == 'structured':
        wl = workloads.StructuredArith(N=width or 1408, chained=args.chained)
        batch = bpg or 1024
        bytes_table, bool_path = BYTES_PER_OP, None
    elif name == 'sha256':
        wl = workloads.Sha256Compress()
        batch = bpg or 4096
        bytes_table, bool_path = BYTES_PER_OP_BOOL, args.bool_path
    else:
        wl = workloads.BoolLayered(W=width or 16384, D=depth or 640,
                                   wiring=os.environ.get('ZKI_C4_WIRING', 'random'))
        batch = bpg or 4096
        bytes_table, bool_path = BYTES_PER_OP_BOOL, args.bool_path
    # ZKI_RANK_BASE (rehearsals only): this launch holds ranks base .. base + world - 1 of a larger job -- a one-GPU box may
    # run at most 6 processes on its card, so the 8 lane shares of BASELINE configs[2] are rehearsed as two launches of 4
    rank_base = int(os.environ.get('ZKI_RANK_BASE', '0'))
    lane_offset = (rank_base + rank) * batch
    ev, inst, wit, n_bad, msgs, host = build_session(zk, name, wl, batch, lane_offset, lane_group, bool_path, args.streams)
    kinds, _, _ = ev.tape()
    # structured: the unit of work is one backend call of the reference's evaluator (every value-returning call it
    # makes for one witness, ladders and scope copies included) -- what the CPU baseline executes call by call
    gates = wl.n_gates if hasattr(wl, 'n_gates') else int((kinds != 9).sum())
    algo_bytes_per_lane = float(sum(bytes_table.get(int(k), 0) * int(c) for k, c in zip(*np.unique(kinds, return_counts=True))))
    info = ev.schedule_info()
    structured = name == 'structured'
    if structured:
        # the bytes of the program that runs (one entry = one gate's reads and write; 13 = the `x != 0` entry a ladder
        # became): the backend calls the rewrite removed -- scope copies, ladder products -- move no bytes at all
        pk = ev.schedule_dump()[0][:, 1] & 0xFF
        algo_bytes_per_lane = float(sum({**BYTES_PER_OP, 13: 64}.get(int(k), 0) * int(c) for k, c in zip(*np.unique(pk, return_counts=True))))
    lds = name in ('c4', 'sha256') and ev.uses_lds_path()
    wide_launches = 1 if lds else info['launches'] - info['sequential_launches']

    counts_t = torch.as_tensor(_DevU64x2(ev.counts_device_ptr()), device='cuda') if dist_on else None
    reduced = torch.zeros(2, dtype=torch.int64, device=red_dev) if dist_on else None

    def step():
        ev.replay()
        ev.synchronize()                # verdict words and counts are final on the engine's stream
        if dist_on:
            reduced.copy_(counts_t)     # 16 bytes out of the engine's counter words
            dist.all_reduce(reduced)    # RCCL over xGMI: {satisfied, failed}
            # the collective runs on RCCL's stream: finish it before the next replay resets the counters
            torch.cuda.current_stream().synchronize()

    for _ in range(warmup):
        step()
    ev.synchronize()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    t0 = time.perf_counter()
    ev_ms = []
    for _ in range(steps):
        step()
        ev_ms.append(ev.last_replay_ms)
    ev.synchronize()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    ranks_seen = None
    if dist_on:
        total = reduced.cpu().tolist()
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # who took part, from the process group itself: every rank's lane share and its own (un-reduced) counts
        own = ev.counts()
        mine = {'rank': rank, 'lane_offset': lane_offset, 'lanes': batch, 'device': ctx['dev_index'],
                'satisfied': int(own[0]), 'failed': int(own[1])}
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, mine)
        ranks_seen = sorted(gathered, key=lambda r: r['rank'])
    else:
        total = list(ev.counts())
    exp_sat = workloads.expected_satisfied(batch * world, rank_base * batch)
    assert total[0] == exp_sat and total[0] + total[1] == batch * world, (total, exp_sat)
    if ranks_seen is not None:
        assert [r['rank'] for r in ranks_seen] == list(range(world)), ranks_seen
        for r in ranks_seen:
            assert r['satisfied'] == workloads.expected_satisfied(r['lanes'], r['lane_offset']), r
    # PCIe-inclusive figure (never `value`): the boundary hands over host buffers every step
    pcie_ms = None
    if world == 1 and headline and not structured and not args.timed_steps_only:
        # page-locked host buffers (what a streaming caller would use): the DMA engine reads them directly
        pin_i = torch.from_numpy(np.ascontiguousarray(inst).reshape(-1)).pin_memory()
        pin_w = torch.from_numpy(np.ascontiguousarray(wit).reshape(-1)).pin_memory()
        ib, wb = pin_i.data_ptr(), pin_w.data_ptr()
        ev.set_inputs(ib, wb, batch)
        ev.replay()
        ev.synchronize()
        tp = time.perf_counter()
        for _ in range(6):   # batch k+1 is handed over while batch k replays (two input sets, copy stream)
            ev.set_inputs(ib, wb, batch)
            ev.replay()
        ev.synchronize()
        pcie_ms = (time.perf_counter() - tp) * 1e3 / 6
        assert list(ev.counts()) == total, (ev.counts(), total)   # the handed-over batches give the resident answer
    first_verdict = None
    if world == 1 and not args.no_first_verdict:
        # relation in -> first verdict out on a fresh session: scheduled at finalize, and streamed (the tape windows are
        # scheduled and uploaded while the later messages are parsed); the headline also with page-locked inputs
        first_verdict = {}
        variants = [('at_finalize', 0, False), ('streamed', 1, False)]
        if name == 'c2' and headline:
            variants.append(('streamed_pinned_inputs', 1, True))
        for key, stream, pinned in variants:
            runs = [first_verdict_seconds(zk, wl, msgs, inst, wit, batch, stream, pinned, bool_path) for _ in range(3 if headline else 2)]
            best = min(runs, key=lambda r: r['total_s'])
            assert best['satisfied'] == exp_sat
            first_verdict[key] = best

    out = None
    if rank == 0:
        ms_per_step = elapsed * 1e3 / steps
        value = gates * batch * world / (elapsed / steps)
        step_kernel_ms = float(np.mean(ev_ms))      # HIP events on the engine's stream around the replay
        fused = info['device_ops'] < len(kinds)
        if name == 'c2':
            metric = 'gate-ops/sec (whole node), 256-bit field, 1M-gate relation, batched witnesses'
            dtype = 'u64x4 (GF(p) Montgomery limbs, exact integer)'
            wl_name = ('BASELINE configs[1]: BN254 scalar field, %d-gate Add/Mul relation (W=%d x D=%d), '
                       'witness batch=%d per GPU, %d GPU(s)' % (gates, wl.W, wl.D, batch, world))
            kernel = 'replay_fused_kernel<8, 0>' if fused else 'replay_kernel<8, false, false>'
            if ev.field_representation() == 2:      # (ZKI_C2_MODULUS: a characteristic the Montgomery kernels do not take)
                kernel = 'replay_generic_kernel (canonical residues, Barrett; unfused)'
                dtype = 'u32 x k (canonical residues, exact integer)'
                wl_name = 'the C2 relation over the characteristic %#x (developer run): ' % wl.p + wl_name
        elif structured:
            metric = ('backend-ops/sec (whole node), 256-bit field, For/Call/Switch relation of ~1M backend calls, batched '
                      'witnesses (one unit = one value-returning ZKBackend call of the reference evaluator)')
            dtype = 'u64x4 (GF(p) Montgomery limbs, exact integer)'
            wl_name = ('structured%s: For over a named function with a nested call and a 2-case Switch, %d iterations, '
                       'BN254 (shape of producers/examples.rs:72-212), witness batch=%d per GPU, %d GPU(s)'
                       % (' (chained: each iteration reads the previous result)' if wl.chained else '', wl.N, batch, world))
            kernel = 'replay_fused_kernel<8, 0> + <8, 1>' if fused else 'replay_kernel<8, false, false>'
        elif name == 'sha256':
            metric = 'gate-ops/sec (whole node), GF(2), SHA-256 compression function (1.2 * 10^5 And/Xor gates in 3,900 levels), batched witnesses'
            dtype = 'u1 (GF(2), %d witnesses per word)' % (32 if lds else 64)
            wl_name = ('SHA-256 compression of one padded block as a Boolean relation (%d gates, %d levels), witness batch=%d per GPU, '
                       '%d GPU(s); expected digests from hashlib' % (gates, info['levels'], batch, world))
            kernel = 'bool_lds_kernel (narrow levels: packets walked by one wave)' if lds else 'bool_replay_kernel'
        else:
            metric = 'gate-ops/sec (whole node), GF(2), 10M-gate And/Xor/Not relation, bit-packed batched witnesses'
            dtype = 'u1 (GF(2), %d witnesses per word)' % (32 if lds else 64)
            wl_name = ('BASELINE configs[3]: GF(2), %d-gate And/Xor/Not relation (W=%d x D=%d), witness batch=%d '
                       'per GPU, %d GPU(s)' % (gates, wl.W, wl.D, batch, world))
            kernel = 'bool_lds_kernel (wire table resident in LDS)' if lds else 'bool_replay_kernel'
        program = {'entries': int(info['device_ops']), 'launches': int(info['launches']), 'batch': batch,
                   'lane_group': lane_group}
        roofline = make_roofline(name, kernel, max(wide_launches, 1), step_kernel_ms, algo_bytes_per_lane * batch, program,
                                 1 if lds else args.streams)
        if structured and 'counters' not in roofline:
            roofline['binding'] = 'as C2 (the same kernels at the same rate per program entry); no counters collected for this program'
            roofline['reading'] = ('%d launches of 1,408 - 8,448 entries; %.2f us per 1000 entry-waves (C2: 259 launches of ~2,500 '
                                   'entries run at about 0.8 us per 1000 entry-waves)'
                                   % (info['launches'], step_kernel_ms * 1e3 / (info['device_ops'] * ((batch + 63) // 64) / 1000.0)))
        n_dev = ctx['n_dev']
        out = {
            'metric': metric,
            'value': value, 'unit': 'backend-ops/s' if structured else 'gate-ops/s', 'n_gpus': world, 'steps': steps,
            'warmup': warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype, 'data': 'synthetic',
            'config': {'workload': wl_name, 'batch_per_gpu': batch,
                       'relation_messages': host['messages'], 'relation_bytes': host['relation_bytes'],
                       'backend_ops_per_witness': int(len(kinds)), 'levels': info['levels'],
                       'launches_per_step': info['launches'], 'sequential_launches': info['sequential_launches'],
                       'wire_table_slots': info['slots'],
                       'program_entries': info['device_ops'],
                       'backend_ops_without_an_entry_of_their_own': int(len(kinds)) - info['device_ops'],
                       'wire_table_MB': round(ev.table_bytes / 1e6, 1), 'lane_group': lane_group,
                       'parallelism': 'witness lanes sharded over %d rank(s) on %d device(s); one all-reduce of 2 x u64 (%s)'
                                      % (world, min(world, n_dev), 'none' if world == 1 else 'RCCL' if ctx['backend'] == 'nccl' else ctx['backend'] + ' rehearsal'),
                       'ranks_seen': ranks_seen, 'rank_base': rank_base,
                       'pcie_inclusive_ms_per_step': None if pcie_ms is None else round(pcie_ms, 3),
                       'satisfied': total[0], 'failed': total[1],
                       'expected_outputs': ('closed form in Python integers' if structured else 'hashlib.sha256 of every lane\'s message' if name == 'sha256' else
                                            'GPU probe pass on another schedule and kernel; %d of %d lanes checked against '
                                            'tests/golden/%s_all_lanes.json (oracle chain)' % (host['lanes_checked_against_golden_hashes'], batch, name)),
                       'host_seconds': {k: round(v, 3) for k, v in host.items() if k.endswith('_s')},
                       'flatten_backend_ops_per_s': round(len(kinds) / max(host['ingest_and_record_s'], 1e-9)),
                       'relation_in_to_first_verdict': first_verdict, 'tape_windows': ev.stream_info()['windows']},
            'roofline': roofline,
        }
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N=1 only
            out['cpu_baseline'] = cpu_baseline(wl, msgs, inst, wit, gates, ev, cpu_budget_s, with_opt=headline)
            if first_verdict:
                out['config']['break_even_batch'] = break_even(min(v['total_s'] for v in first_verdict.values()), ms_per_step, batch,
                                                               out['cpu_baseline'])
    ev.close()

    if rank == 0 and world == 1 and name == 'c4' and lds and not width and not bpg and not args.timed_steps_only:
        # the LDS-resident kernel gives every 32-witness slice a workgroup of its own that walks the whole program: the
        # BASELINE batch of 4096 fills 128 of the 256 CUs.  The same program over 8192 witnesses fills the chip.
        vb = 8192
        vev, _, _, vbad, _, vhost = build_session(zk, name, wl, vb, 0, 0, bool_path, args.streams)
        for _ in range(2):
            vev.replay()
        vev.synchronize()
        vms = []
        for _ in range(10):
            vev.replay()
            vev.synchronize()
            vms.append(vev.last_replay_ms)
        assert list(vev.counts()) == [workloads.expected_satisfied(vb), vbad]
        vk = float(np.mean(vms))
        out['batch_8192'] = {'batch': vb, 'workgroups': vb // 32, 'ms_per_step': vk, 'value': gates * vb / (vk * 1e-3), 'unit': 'gate-ops/s',
                             'lanes_checked_against_golden_hashes': vhost['lanes_checked_against_golden_hashes'],
                             'what': 'same relation, 8192 witnesses = 256 workgroups, one per CU; ms_per_step is the HIP-event time of '
                                     'the replay (input packing + the LDS-resident kernel + the verdict reduction)'}
        vev.close()

    if rank == 0 and world == 1 and name == 'c2' and headline and not args.no_hbm_variant and not width and not bpg:
        # the same relation with 4096 witnesses replayed at once: a 1.05 GB wire table cannot sit in the 256 MiB
        # Infinity Cache, so here the memory side IS HBM (the headline batch of 1024 keeps its 263 MB table on-die)
        hb = 4096
        hev, _, _, hbad, _, hhost = build_session(zk, name, wl, hb, 0, hb, None, args.streams)
        for _ in range(2):
            hev.replay()
        hev.synchronize()
        hms = []
        for _ in range(8):
            hev.replay()
            hev.synchronize()
            hms.append(hev.last_replay_ms)
        assert list(hev.counts()) == [workloads.expected_satisfied(hb), hbad]
        hinfo = hev.schedule_info()
        hk = float(np.mean(hms))
        rh = make_roofline('c2_hbm', out['roofline']['kernel'], hinfo['launches'] - hinfo['sequential_launches'], hk,
                           algo_bytes_per_lane * hb, {'entries': int(hinfo['device_ops']), 'launches': int(hinfo['launches']),
                                                      'batch': hb, 'lane_group': hb}, args.streams)
        rh.update({'batch': hb, 'lane_group': hb, 'wire_table_MB': round(hev.table_bytes / 1e6, 1), 'ms_per_step': hk,
                   'value': gates * hb / (hk * 1e-3), 'lanes_checked_against_golden_hashes': hhost['lanes_checked_against_golden_hashes'],
                   'what': 'same program, all 4096 witnesses in flight (one lane group): the working set is 4x the Infinity '
                           'Cache, so the memory side is HBM; ms_per_step is the HIP-event time of the replay'})
        out['roofline_hbm'] = rh
        hev.close()
    return out


_RESULT_FD = None


def keep_stdout_for_the_result():
    """The contract is ONE JSON line on stdout.  Libraries write there too (gloo announces its peers on stdout, RCCL can
    be told to log): from here on file descriptor 1 goes to stderr, and only emit_json_line() writes to the real stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit_json_line(out):
    line = (json.dumps(out) + '\n').encode()
    if _RESULT_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, line)


def launch_ranks(n):
    """Start one child process per rank (never exec: this process stays a plain parent and has not touched the
    GPU) and return the first non-zero exit code, or 0."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:      # a rank died: the others would wait in a collective for ever
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=['c2', 'c4', 'c5', 'structured', 'sha256'], default='c2',
                    help='c2 = BASELINE configs[1] (headline); c4 = GF(2) 10M-gate relation, batch 4096; c5 = R1CS rows; '
                         'structured = For / Call / Switch relation of ~1M backend calls (--width = loop iterations)')
    ap.add_argument('--batch-per-gpu', type=int, default=0)
    ap.add_argument('--width', type=int, default=0)
    ap.add_argument('--depth', type=int, default=0)
    ap.add_argument('--bool-path', choices=['auto', 'hbm', 'lds'], default='auto')
    ap.add_argument('--streams', type=int, default=2, help='lane halves replayed concurrently on this many HIP streams')
    ap.add_argument('--lane-group', type=int, default=0, help='replay lane groups of this size one after the other')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-hbm-variant', action='store_true', help='c2: skip the 4096-witnesses-in-flight variant')
    ap.add_argument('--chained', action='store_true', help='structured: every iteration depends on the one before')
    ap.add_argument('--coefs', default='random', choices=['random', 'small'],
                    help='c5: random field elements (BASELINE configs[4]) or 1 / -1 / 16-bit signed integers (the coefficient classes)')
    ap.add_argument('--no-first-verdict', action='store_true', help='c2: skip the relation-in -> first-verdict-out sessions')
    ap.add_argument('--no-secondary', action='store_true', help='c2 on one GPU: skip the c4 / c5 / structured lines')
    ap.add_argument('--full-line', action='store_true', help='print everything on the line (default: the compact line; the rest in the detail file)')
    ap.add_argument('--timed-steps-only', action='store_true',
                    help='nothing but the probe, the warm-up and the timed steps (what a rocprofv3 trace of the timed region needs): '
                         '= --no-cpu-baseline --no-hbm-variant --no-first-verdict --no-secondary and no PCIe-inclusive passes')
    args = ap.parse_args()
    if args.timed_steps_only:
        args.no_cpu_baseline = args.no_hbm_variant = args.no_first_verdict = args.no_secondary = True

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args.gpus))   # nothing in this process has touched torch or the GPU

    keep_stdout_for_the_result()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    args.gpus = world
    if os.environ.get('ZKI_BENCH_FAIL_RANK') == str(rank) and world > 1:
        sys.exit('rank %d told to fail (launcher test)' % rank)
    n_dev = torch.cuda.device_count()
    if n_dev == 0 or not torch.cuda.is_available():
        sys.exit('bench.py needs a GPU: the replay path has no CPU fallback')
    # One process per GPU, counts reduced by RCCL.  With fewer devices than ranks (a one-GPU box) the ranks share
    # the card and the reduction runs over gloo on CPU tensors: a rehearsal of the same sharding, said so in the line.
    backend = os.environ.get('ZKI_DIST_BACKEND', 'nccl' if n_dev >= world else 'gloo')
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    red_dev = 'cuda' if backend == 'nccl' else 'cpu'
    # ZKI_FORCE_DIST=1 with one rank: the process group, the view of the engine's device counters and the all-reduce run
    # exactly as they do with N ranks (the GPU tier uses it to exercise the RCCL path on a one-GPU box)
    dist_on = world > 1 or os.environ.get('ZKI_FORCE_DIST') == '1'
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend)

    zk = entry.ensure_built()   # rebuilds when any source differs from what lib/libzkgpu.so was built from
    from zkinterface_ir_amd import workloads
    ctx = {'world': world, 'rank': rank, 'dist': dist, 'torch': torch, 'red_dev': red_dev, 'dist_on': dist_on,
           'backend': backend, 'n_dev': n_dev, 'dev_index': dev_index}

    if args.workload == 'c5':
        out = bench_c5(args, zk, workloads, ctx, args.steps, args.warmup, coef_kind=args.coefs)
    else:
        out = bench_tape(args.workload, args, zk, workloads, ctx, args.steps, args.warmup, True)
    default_c2 = args.workload == 'c2' and not (args.width or args.depth or args.batch_per_gpu or args.lane_group)
    if rank == 0 and world == 1 and not dist_on and default_c2 and not args.no_secondary:
        # the other workloads, measured in this same invocation with bounded CPU samples (a few seconds each)
        t0 = time.time()
        sec = {}
        sec['c4'] = bench_tape('c4', args, zk, workloads, ctx, 20, 3, False, cpu_budget_s=4.0)
        sec['c5'] = bench_c5(args, zk, workloads, ctx, 5, 1, cpu_budget_s=4.0)
        sec['c5_small'] = bench_c5(args, zk, workloads, ctx, 5, 1, cpu_budget_s=0, coef_kind='small')
        sec['structured'] = bench_tape('structured', args, zk, workloads, ctx, 20, 3, False, cpu_budget_s=4.0)
        # the same calls as a dependency chain (every iteration reads the previous result): strands
        args.chained = True
        sec['structured_chained'] = bench_tape('structured', args, zk, workloads, ctx, 10, 2, False, cpu_budget_s=2.0)
        args.chained = False
        # a real Boolean circuit: thousands of narrow levels
        sec['sha256'] = bench_tape('sha256', args, zk, workloads, ctx, 10, 2, False, cpu_budget_s=2.0)
        out['secondary'] = sec
        out['secondary_wall_s'] = round(time.time() - t0, 1)
    if rank == 0:
        # the line carries the numbers; everything else (prose, counter lists, per-stage host times, the sessions behind
        # first_verdict_s) goes to the detail file named in it
        detail = os.environ.get('ZKI_BENCH_DETAIL', os.path.join('profiles', 'bench_detail_latest.json'))
        try:
            with open(os.path.join(ROOT, detail), 'w') as f:
                json.dump(out, f, indent=1)
        except OSError:
            detail = None
        emit_json_line(out if args.full_line else compact_line(out, detail))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
