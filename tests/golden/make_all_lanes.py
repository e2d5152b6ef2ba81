#!/usr/bin/env python3
"""Writes tests/golden/c2_all_lanes.json and c4_all_lanes.json: one 64-bit hash of the 64 output wires for EVERY lane
of the full-size BASELINE workloads -- 8192 lanes of C2 (configs[1] = lanes 0..1023, configs[2] = 8 ranks x 1024) and
4096 lanes of C4 -- so that the GPU tier and bench.py never take an expected output on trust.

The chain: the literal oracle (oracle/zki_oracle.cpp, restatement of rust/src/consumers/evaluator.rs) made the six-lane
digests of c2_digests.json / c4_digests.json; here the fast checkers (cpu_opt for C2, the numpy bit-sliced checker for
C4) are first asserted to reproduce those six lanes, then run over all lanes.

  python tests/golden/make_all_lanes.py        (C2: 8192 lanes of a 2^20-gate relation, a few minutes on 8 cores)"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import cpu_checkers  # noqa: E402
from zkinterface_ir_amd import workloads  # noqa: E402


def c2():
    wl = workloads.ArithLayered()
    six = json.load(open(os.path.join(HERE, 'c2_digests.json')))
    hashes = []
    for rank in range(8):                       # lane_offset = rank * 1024, as bench.py --gpus N shards the batch
        inst, wit = wl.inputs(1024, rank * 1024)
        out = cpu_checkers.arith_layered_outputs(wl, inst, wit)
        if rank == 0:
            for lane, want in six['lanes'].items():
                vals = [int.from_bytes(out[int(lane), t].tobytes(), 'little') for t in range(wl.n_out)]
                assert hashlib.sha256('\n'.join(str(v) for v in vals).encode()).hexdigest() == want['sha256'], lane
            print('cpu_opt reproduces the six oracle lanes of c2_digests.json')
        hashes += [cpu_checkers.lane_hash(out[lane]) for lane in range(1024)]
        print('c2 rank', rank, hashes[-1])
    fx = {'workload': 'ArithLayered(W=4096, D=256, seed=0x5EED0001) over BN254 r; lane g = lane g %% 1024 of inputs(1024, '
                      'lane_offset = g // 1024 * 1024)',
          'hash': 'first 16 hex digits of sha256 over the 64 output wires as 32-byte little-endian values',
          'made_by': 'cpu_opt (oracle/cpu_opt.cpp), pinned to the literal oracle on the lanes of c2_digests.json',
          'hashes': hashes}
    json.dump(fx, open(os.path.join(HERE, 'c2_all_lanes.json'), 'w'))


def c4():
    wl = workloads.BoolLayered()
    six = json.load(open(os.path.join(HERE, 'c4_digests.json')))
    inst, wit = wl.inputs(4096)
    out = cpu_checkers.bool_layered_outputs(wl, inst, wit)
    for lane, want in six['lanes'].items():
        assert cpu_checkers.bool_digest(out[int(lane)]) == want['sha256'], lane
    print('the numpy checker reproduces the six oracle lanes of c4_digests.json')
    fx = {'workload': 'BoolLayered(W=16384, D=640, seed=0xB001C4) over GF(2), inputs(4096)',
          'hash': 'first 16 hex digits of sha256 over the 64 output bits, one byte each',
          'made_by': 'tests/cpu_checkers.py bool_layered_outputs, pinned to the literal oracle on the lanes of c4_digests.json',
          'hashes': [cpu_checkers.lane_hash(out[lane]) for lane in range(4096)]}
    json.dump(fx, open(os.path.join(HERE, 'c4_all_lanes.json'), 'w'))


if __name__ == '__main__':
    which = sys.argv[1:] or ['c4', 'c2']
    if 'c4' in which:
        c4()
    if 'c2' in which:
        c2()
