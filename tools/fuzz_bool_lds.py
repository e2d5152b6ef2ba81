#!/usr/bin/env python3
"""Randomised shapes for the LDS-resident GF(2) kernel against the numpy checker (tests/cpu_checkers.py): widths, depths,
ragged batches, block sizes forced and free.  `python tools/fuzz_bool_lds.py [n_cases] [seed]` on a GPU box; tests/test_full_size.py runs 20 cases of it in the GPU tier."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as entry  # noqa: E402

zk = entry.load_package()
import cpu_checkers  # noqa: E402
from zkinterface_ir_amd import workloads  # noqa: E402


def run(n_cases=24, seed=1, verbose=True):
    """-> number of cases run; raises AssertionError on the first mismatch"""
    rng = np.random.default_rng(seed)
    saved = os.environ.get('ZKGPU_LDS_BLOCK_ROWS')
    try:
        for case in range(n_cases):
            W = int(rng.choice([96, 700, 2048, 2049, 4097, 6000, 9000, 12288, 16384, 19000]))
            D = int(rng.integers(1, 9))
            n_out = int(rng.integers(1, min(W, 64) + 1))
            n_inst0 = int(rng.integers(1, min(W - 1, 200) + 1))
            batch = int(rng.choice([1, 31, 32, 33, 64, 100, 257]))
            br = int(rng.choice([0, 0, 4, 6, 8, 9, 10, 12]))
            if br:
                os.environ['ZKGPU_LDS_BLOCK_ROWS'] = str(br)
            else:
                os.environ.pop('ZKGPU_LDS_BLOCK_ROWS', None)
            mix = [(45, 45), (45, 45), (100, 0), (0, 100), (0, 0), (97, 3), (3, 90), (50, 50)][int(rng.integers(0, 8))]
            wl = workloads.BoolLayered(W=W, D=D, n_instance0=n_inst0, n_out=n_out, seed=int(rng.integers(1, 1 << 30)), mix=mix)
            inst, wit = wl.inputs(batch)
            outs = cpu_checkers.bool_layered_outputs(wl, inst, wit)
            inst = inst.copy()
            wl.set_expected_outputs(inst, outs, corrupt_every=0)
            want = np.full(batch, zk.NO_FAIL, dtype=np.uint32)
            for lane in range(0, batch, 3):
                inst[lane, wl.n_instance0 + lane % n_out, 0] ^= 1
                want[lane] = lane % n_out
            ev = zk.Evaluator()
            ev.set_option('bool_path', 'lds')
            if case % 3 == 2:
                ev.set_option('stream', '3000')   # windows scheduled while the messages come in
            ev.declare_inputs(wl.n_instance, wl.n_witness)
            for m in wl.relation_messages():
                ev.ingest_message(m)
            ev.finalize()
            ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
            ev.replay()
            ev.synchronize()
            first, flags = ev.lane_results(batch)
            ok = np.array_equal(first, want) and not flags.any()
            if verbose:
                print('case %2d W %5d D %d out %2d batch %3d block_rows %2d mix %s -> %s' % (case, W, D, n_out, batch, br, mix, 'ok' if ok else 'MISMATCH'), flush=True)
            assert ok, 'case %d (seed %d): W %d D %d n_out %d batch %d block_rows %d mix %s' % (case, seed, W, D, n_out, batch, br, mix)
            ev.close()
    finally:
        if saved is None:
            os.environ.pop('ZKGPU_LDS_BLOCK_ROWS', None)
        else:
            os.environ['ZKGPU_LDS_BLOCK_ROWS'] = saved
    return n_cases


if __name__ == '__main__':
    n = run(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print('all %d cases agree with the CPU checker' % n)
