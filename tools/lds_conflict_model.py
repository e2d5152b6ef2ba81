#!/usr/bin/env python3
"""LDS bank-conflict model of a GF(2) program for the LDS-resident kernel (host work only, no GPU):

  python tools/lds_conflict_model.py [W] [D]        (default: the C4 relation, 16384 x 640)

For every row of the program and every operand read of a wave (ds_read_b32: 64 lanes served as two groups of 32 lanes,
bank = slot mod 32, one LDS-array cycle per distinct address on the busiest bank; MI355X_MICROARCH.md, LDS) the cycles
the read takes; prints the mean per group for the four reads (1.0 = conflict-free) and the time the schedule took."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def group_cycles(slots):
    """slots: [n_groups, 32] -> cycles per group = max over banks of the distinct addresses on it"""
    bank = slots & 31
    order = np.lexsort((slots, bank))            # per row: sort by (bank, slot)
    rows = np.arange(slots.shape[0])[:, None]
    s, b = slots[rows, order], bank[rows, order]
    new = np.ones_like(s, dtype=bool)
    new[:, 1:] = (s[:, 1:] != s[:, :-1]) | (b[:, 1:] != b[:, :-1])
    cycles = np.zeros(slots.shape[0], dtype=np.int64)
    for k in range(32):
        cycles = np.maximum(cycles, ((b == k) & new).sum(axis=1))
    return cycles


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    D = int(sys.argv[2]) if len(sys.argv) > 2 else 640
    zk = entry.load_package()
    from zkinterface_ir_amd import workloads
    wl = workloads.BoolLayered(W=W, D=D)
    ev = zk.Evaluator()
    for k, v in (('ZKI_BANK_AWARE', 'bank_aware'), ('ZKI_SCHED_THREADS', 'schedule_threads')):
        if os.environ.get(k):
            ev.set_option(v, os.environ[k])
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in wl.relation_messages():
        ev.ingest_message(m)
    t0 = time.time()
    ev.finalize()
    t1 = time.time()
    P = ev.lds_program(0)
    rows = np.asarray(P['rows'], dtype=np.uint16)
    n_real = (len(rows) // 3 // 2048) * 2048 - P['block_rows'] * 2048   # without the slack rows behind the last block
    r = rows[:3 * n_real].reshape(-1, 3).astype(np.int64)   # [op][dst, a, b]
    out = {}
    for name, col, par in (('a0', 1, 0), ('b0', 2, 0), ('a1', 1, 1), ('b1', 2, 1)):
        v = r[par::2, col].reshape(-1, 32)       # even / odd ops: 32 consecutive threads = one lane group
        out[name] = float(group_cycles(v).mean())
    print('schedule %.2f s | slots %d | rows %d | mean LDS cycles per 32-lane read group: %s'
          % (t1 - t0, ev.schedule_info()['slots'], n_real // 2048, ' '.join('%s %.3f' % kv for kv in out.items())))


if __name__ == '__main__':
    main()
