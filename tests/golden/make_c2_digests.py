#!/usr/bin/env python3
"""Writes tests/golden/c2_digests.json: for a few lanes of the BASELINE configs[1] workload (BN254, W=4096 x D=256
Add/Mul relation, workloads.ArithLayered defaults) the SHA-256 of the 64 output-wire values as the oracle
(oracle/zki_oracle.cpp, the CPU restatement of rust/src/consumers/evaluator.rs) computes them.  The GPU tier compares
the replayed values of the same lanes with these digests, so the full-size result is pinned across rounds.

  python tests/golden/make_c2_digests.py        (about a second per lane)"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
from helpers import oracle_lane  # noqa: E402
from zkinterface_ir_amd import workloads  # noqa: E402

LANES = [0, 1, 96, 97, 513, 1023]


def main():
    wl = workloads.ArithLayered()
    msgs = wl.relation_messages(with_epilogue=False, free_last=False)
    inst, wit = wl.inputs(1024)
    out = {'workload': 'ArithLayered(W=4096, D=256, seed=0x5EED0001) over BN254 r', 'n_out': wl.n_out,
           'digest': 'sha256 of the decimal output-wire values joined by newlines', 'lanes': {}}
    for lane in LANES:
        iv = [int.from_bytes(inst[lane, k].tobytes(), 'little') for k in range(wl.n_instance0)]
        wv = [int.from_bytes(wit[lane, k].tobytes(), 'little') for k in range(wl.n_witness)]
        run = oracle_lane(wl.mod_le, iv, wv, msgs, wl.width, trace=False)
        assert run.violations == []
        vals = [run.get(w) for w in wl.output_wire_ids()]
        out['lanes'][str(lane)] = {'sha256': hashlib.sha256('\n'.join(str(v) for v in vals).encode()).hexdigest(),
                                   'first_output': str(vals[0])}
        print(lane, out['lanes'][str(lane)]['sha256'])
    with open(os.path.join(HERE, 'c2_digests.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
