#!/usr/bin/env python3
"""Writes tests/golden/c5_digests.json: for a few lanes of the BASELINE configs[4] workload (2^20-row R1CS over
BN254, workloads.R1csSynthetic defaults, batch 1024) the values of 257 product-row variables z_r = <a,w>*<b,w> mod p
spread over all dependency levels (the last one included), computed with plain Python integers from the
mathematical definition of the rows.  (The reference holds no row checker of its own -- SURVEY.md 8c: the zkinterface
Simulator is a crates.io dependency -- so Python's arbitrary-precision integers are the anchor here.)  The GPU tier
compares the variables that zkgpu_r1cs_assign generated for the same lanes with these digests.

  python tests/golden/make_c5_digests.py        (about 10 s per lane)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import __graft_entry__  # noqa: E402

__graft_entry__.load_package()
import cpu_checkers  # noqa: E402
from zkinterface_ir_amd import workloads  # noqa: E402

LANES = [0, 1, 97, 1023]


def main():
    wl = workloads.R1csSynthetic()
    w = wl.witnesses(1024)
    sample = cpu_checkers.r1cs_sample_vars(wl)
    out = {'workload': 'R1csSynthetic(M=2^20, n_base=4096, n_coefs=2^16, seed=0xC5) over BN254 r, lanes of the '
                       'batch-1024 witness set', 'sampled_variables': len(sample),
           'digest': 'sha256 of the sampled variable values as 32-byte little-endian strings (cpu_checkers.r1cs_sample_vars)',
           'lanes': {}}
    for lane in LANES:
        val = cpu_checkers.r1cs_lane_assignment(wl, w[lane])
        out['lanes'][str(lane)] = {'sha256': cpu_checkers.r1cs_digest([val[v] for v in sample], wl.width),
                                   'last_z': str(val[wl.last_z])}
        print(lane, out['lanes'][str(lane)])
    with open(os.path.join(HERE, 'c5_digests.json'), 'w') as f:
        json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
