#!/usr/bin/env python3
"""Timing experiments on the LDS-resident GF(2) kernel at the C4 shape (C4_WIDTH / C4_DEPTH / C4_BATCH change it).
ZKGPU_VARIANT=<name> loads a library built by tools/build_variant.sh (extra -D flags for kernels_bool.hip), ZKGPU_LDS_BLOCK_ROWS
forces the block size the engine would otherwise pick.  The expected outputs are not filled in, so every lane fails
at its first non-zero output bit: the checksum of the first-fail words printed with the time is the same for every
variant that computes the right values (profiles/r02_tuning_sweeps.txt holds the numbers this produced)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

zk = entry.load_package()
if os.environ.get('ZKGPU_VARIANT'):   # a library built by tools/build_variant.sh
    zk.LIB_PATH = os.path.join(ROOT, 'zkinterface-ir_amd', 'lib', 'variants', 'libzkgpu_%s.so' % os.environ['ZKGPU_VARIANT'])
from zkinterface_ir_amd import workloads  # noqa: E402

wl = workloads.BoolLayered(W=int(os.environ.get('C4_WIDTH', 16384)), D=int(os.environ.get('C4_DEPTH', 640)))
batch = int(os.environ.get('C4_BATCH', 4096))
inst, wit = wl.inputs(batch)
ev = zk.Evaluator()
ev.declare_inputs(wl.n_instance, wl.n_witness)
for m in wl.relation_messages():
    ev.ingest_message(m)
ev.finalize()
ev.set_inputs(inst.tobytes(), wit.tobytes(), batch)
ms = []
for k in range(12):
    ev.replay()
    ev.synchronize()
    ms.append(ev.last_replay_ms)
import zlib  # noqa: E402
# the expected outputs are not filled in, so every lane fails at its first non-zero output bit: the checksum of the
# first-fail words is the same for every variant that computes the right values
print('W %d D %d batch %d' % (wl.W, wl.D, batch), 'variant %-6s first-fail crc %08x  replay ms: min %.3f median %.3f' % (os.environ.get('ZKGPU_VARIANT', '-'), zlib.crc32(ev.lane_results(batch)[0].tobytes()), min(ms[2:]), float(np.median(ms[2:]))))
