#!/usr/bin/env python3
"""Soak: fresh seeds of the random structured relations of the test tier (tests/random_circuits.py: For / Call / Switch /
anonymous functions over eleven fields incl. GF(2), GF(3), 65537, P512) through the default schedule on the GPU -- strands
with LDS values, operand copies, joined levels, entry prefetch -- and, every other seed, a streamed ingest; violations of
sampled lanes, the counts and (where the statement holds) the surviving wires against the oracle.

  python tools/soak_random_relations.py [first_seed] [n_seeds]        (on an MI355X; about 0.15 s per seed)

Not part of the test tiers (they run 20-45 seeds of the same generator); a one-off record goes to profiles/."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import __graft_entry__ as entry  # noqa: E402

zk = entry.load_package()
from helpers import batch_arrays, oracle_lane  # noqa: E402
from random_circuits import Gen  # noqa: E402
from test_fuzz_host import FIELDS  # noqa: E402


def one(seed):
    p, boolean = FIELDS[seed % len(FIELDS)]
    g = Gen(seed, p, boolean)
    rel, mod_le = g.relation(n_top=10 + seed % 9)
    lanes = 66
    rows_i, rows_w = g.lane_inputs(lanes, seed + 1000)
    ev = zk.Evaluator()
    if seed % 2:
        ev.set_option('stream', '32')
    ev.declare_inputs(g.n_inst, g.n_wit)
    ev.ingest_message(rel)
    ev.finalize()
    info = ev.schedule_info()
    inst, wit = batch_arrays(rows_i, rows_w, ev.elem_bytes)
    ev.set_inputs(inst if g.n_inst else None, wit if g.n_wit else None, lanes)
    ev.replay()
    ev.synchronize()
    for lane in range(0, lanes, 5):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32, trace=False)
        assert ev.get_violations(lane) == ref.violations, (seed, lane, ev.get_violations(lane), ref.violations)
    first, _ = ev.lane_results(lanes)
    n_ok = int((first == zk.NO_FAIL).sum())
    assert ev.counts() == (n_ok, lanes - n_ok), seed
    ref = oracle_lane(mod_le, rows_i[0], rows_w[0], [rel], 32, trace=False)
    if not ref.violations:
        for wid in range(0, 40):
            want, got = ref.get(wid), ev.get(wid, lanes)
            assert (want is None) == (got is None), (seed, wid)
            if want is not None:
                assert got[0] == want, (seed, wid)
    ev.close()
    return info['sequential_launches'], info['launches']


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    t0 = time.time()
    strands = launches = 0
    for seed in range(first, first + n):
        s, l = one(seed)
        strands += s
        launches += l
        if (seed - first) % 100 == 99:
            print('%d seeds, %.0f s' % (seed - first + 1, time.time() - t0), flush=True)
    print('%d random relations (seeds %d..%d), %d launches of which %d strands: every sampled lane, count and surviving wire '
          'equals the oracle (%.0f s)' % (n, first, first + n - 1, launches, strands, time.time() - t0))


if __name__ == '__main__':
    main()
