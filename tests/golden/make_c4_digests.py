#!/usr/bin/env python3
"""Writes tests/golden/c4_digests.json: for a few lanes of the BASELINE configs[3] workload (GF(2), W=16384 x D=640
And/Xor/Not relation, workloads.BoolLayered defaults, batch 4096) the SHA-256 of the 64 output bits as the oracle
(oracle/zki_oracle.cpp, the CPU restatement of rust/src/consumers/evaluator.rs:924-938) computes them, one
reference-style Evaluator run per lane.  The GPU tier compares both GF(2) kernels (LDS-resident and HBM table) with
these digests, and the numpy checker of tests/cpu_checkers.py -- which then vouches for all 4096 lanes -- is pinned
to them too.

  python tests/golden/make_c4_digests.py        (about 5 s per lane)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import cpu_checkers  # noqa: E402
from helpers import oracle_lane  # noqa: E402
from zkinterface_ir_amd import workloads  # noqa: E402

LANES = [0, 1, 96, 97, 2049, 4095]


def main():
    wl = workloads.BoolLayered()
    msgs = wl.relation_messages(with_epilogue=False, free_last=False)
    inst, wit = wl.inputs(4096)
    out = {'workload': 'BoolLayered(W=16384, D=640, seed=0xB001C4) over GF(2), lanes of the batch-4096 input set',
           'n_out': wl.n_out, 'digest': "sha256 of the output bits as a string of '0'/'1'", 'lanes': {}}
    for lane in LANES:
        run = oracle_lane(wl.mod_le, inst[lane, :wl.n_instance0, 0].tolist(), wit[lane, :, 0].tolist(), msgs, 1, trace=False)
        assert run.violations == []
        bits = [run.get(w) for w in wl.output_wire_ids()]
        assert all(b in (0, 1) for b in bits)
        out['lanes'][str(lane)] = {'sha256': cpu_checkers.bool_digest(bits), 'bits': ''.join(str(b) for b in bits)}
        print(lane, out['lanes'][str(lane)])
    with open(os.path.join(HERE, 'c4_digests.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
