#!/usr/bin/env python3
"""Host side of the C4 relation (10.5 M gates over GF(2)) without a GPU: ingest + record, then zkgpu_finalize with the
scheduler's stage times (ZKI_SCHED_PROFILE=1) for a few thread counts.  KEEP=1 keeps the message buffers alive while the
relation is scheduled (what bench.py does)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import __graft_entry__ as entry  # noqa: E402

zk = entry.load_package()
from zkinterface_ir_amd import workloads  # noqa: E402

wl = workloads.BoolLayered(W=int(os.environ.get('C4_WIDTH', 16384)), D=int(os.environ.get('C4_DEPTH', 640)))
msgs = [b"".join(wl.relation_messages())] if os.environ.get("ONE_BUFFER") else wl.relation_messages()
for threads in [int(t) for t in os.environ.get('THREADS', '8').split(',')]:
    ev = zk.Evaluator()
    ev.set_option('schedule_threads', str(threads))
    if os.environ.get('STREAM'):
        ev.set_option('stream', os.environ['STREAM'])
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    t0 = time.time()
    for m in msgs:
        ev.ingest_message(m)
    t1 = time.time()
    ev.finalize()
    t2 = time.time()
    print('threads %d: ingest + record %.3f s, finalize %.3f s, %s' % (threads, t1 - t0, t2 - t1, ev.schedule_info()), flush=True)
    ev.close()
