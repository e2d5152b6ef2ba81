"""GPU tier: the product's N > 1 path, started the way the driver starts it (`python bench.py --gpus N`, no external
launcher).  On a one-GPU box the ranks share the card and the count reduction runs over gloo (bench.py picks that
itself when fewer devices than ranks are visible): same lane sharding, same per-rank Engine, same reduction of the
engine's device counters as the 8-GPU run over RCCL (BASELINE configs[2])."""
import json
import os
import subprocess
import sys

import pytest

from zkinterface_ir_amd import workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=600):
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + [str(a) for a in args], env=env,
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout      # stdout carries ONE line: rank 0's JSON
    # (library chatter -- gloo announces its peers on stdout -- is sent to stderr)
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_launches_two_ranks_itself_and_reduces_the_counts():
    r = run_bench('--gpus', 2, '--steps', 2, '--warmup', 1, '--batch-per-gpu', 256, '--width', 512, '--depth', 8,
                  '--no-cpu-baseline')
    assert r['n_gpus'] == 2 and r['scaling'] == 'weak'
    # lanes 0..511 of the global batch: corrupted 0, 97, 194, 291, 388, 485 -- 3 on each rank
    assert r['config']['satisfied'] == workloads.expected_satisfied(512) == 506
    assert r['config']['failed'] == 6
    assert r['value'] > 0 and r['roofline']['frac'] > 0


@pytest.mark.gpu
def test_three_ranks_uneven_corruption_boolean_and_r1cs():
    r = run_bench('--gpus', 3, '--workload', 'c4', '--steps', 1, '--warmup', 1, '--batch-per-gpu', 128, '--width', 2048,
                  '--depth', 4, '--no-cpu-baseline')
    assert r['n_gpus'] == 3
    assert r['config']['satisfied'] == workloads.expected_satisfied(384) == 380   # 0, 97 | 194 | 291
    r = run_bench('--gpus', 2, '--workload', 'c5', '--steps', 1, '--warmup', 1, '--batch-per-gpu', 128, '--width', 2048,
                  '--no-cpu-baseline')
    assert r['n_gpus'] == 2 and r['config']['satisfied'] == workloads.expected_satisfied(256) == 253


@pytest.mark.gpu
def test_a_failing_rank_fails_the_launcher():
    env = dict(os.environ, ZKI_BENCH_FAIL_RANK='1')
    env.pop('WORLD_SIZE', None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0',
                          '--batch-per-gpu', '64', '--width', '64', '--depth', '3', '--no-cpu-baseline'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0


@pytest.mark.gpu
def test_rccl_path_with_one_rank():
    """what the 8-GPU run does per step -- process group over RCCL, a torch view of the engine's device counters, the
    all-reduce, the max-over-ranks timing -- with a world of one rank, so that the code path the driver launches on a
    multi-GPU node has at least run against the real RCCL on this box (ZKI_FORCE_DIST=1)"""
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               ZKI_FORCE_DIST='1')
    env.pop('ZKI_DIST_BACKEND', None)
    for extra in ([], ['--workload', 'c5', '--width', '2048'], ['--workload', 'c4', '--width', '2048', '--depth', '4']):
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                              '--batch-per-gpu', '128', '--no-cpu-baseline', '--no-hbm-variant', '--no-first-verdict']
                             + (extra or ['--width', '256', '--depth', '6']), env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        r = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][0])
        assert r['n_gpus'] == 1 and r['config']['satisfied'] == workloads.expected_satisfied(128) == 126


@pytest.mark.gpu
def test_configs2_lane_shares_at_full_size_in_two_launches_of_four_ranks():
    """BASELINE configs[2]: the full 2^20-gate BN254 relation, 8 ranks x 1024 witnesses.  The pool allows at most 6
    processes on one card, so the 8 lane shares are rehearsed as two launches of 4 ranks (ZKI_RANK_BASE = 0 and 4), each
    rank with its own engine, 263 MB wire table and 1024-lane share at its global lane offset; every rank's probe outputs
    are checked against tests/golden/c2_all_lanes.json (all 8192 lanes) inside bench.py, the counts are reduced over the
    process group, and `ranks_seen` comes from an all_gather over it.  Together: 8192 - 85 satisfied."""
    sat = 0
    for base in (0, 4):
        os.environ['ZKI_RANK_BASE'] = str(base)
        try:
            r = run_bench('--gpus', 4, '--steps', 2, '--warmup', 1, '--no-cpu-baseline', timeout=900)
        finally:
            del os.environ['ZKI_RANK_BASE']
        assert r['n_gpus'] == 4 and r['config']['rank_base'] == base and r['config']['batch_per_gpu'] == 1024
        assert 'BASELINE configs[1]' in r['config']['workload'] and '1048576-gate' in r['config']['workload']
        seen = r['config']['ranks_seen']
        assert [s['rank'] for s in seen] == [0, 1, 2, 3]
        assert [s['lane_offset'] for s in seen] == [(base + k) * 1024 for k in range(4)]
        for s in seen:
            assert s['lanes'] == 1024 and s['satisfied'] == workloads.expected_satisfied(1024, s['lane_offset'])
            assert s['satisfied'] + s['failed'] == 1024
        assert r['config']['satisfied'] == sum(s['satisfied'] for s in seen) == workloads.expected_satisfied(4096, base * 1024)
        assert '1024 of 1024 lanes checked' in r['config']['expected_outputs']
        sat += r['config']['satisfied']
    assert sat == 8192 - 85
