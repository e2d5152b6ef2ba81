#!/usr/bin/env python3
"""Where a strand's time goes, level by level (developer instrumentation):

  (cd zkinterface-ir_amd && touch csrc/kernels_arith.hip && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DZKGPU_STRAND_STAMPS")
  ZKGPU_STRAND_STAMPS=/tmp/stamps.bin python bench.py --workload structured --chained --no-cpu-baseline --no-first-verdict --steps 2 --warmup 1
  python tools/strand_stamps.py /tmp/stamps.bin [levels to print]

(a developer build: the stamps are compiled out of the product library.)  The strand kernel of lane block 0 stamps, per level and wave: level start, entry fetched (scalar load back), entry done
(all its loads / stores back), barrier passed.  Cycle counter of the shader clock."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 4).astype(np.int64)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
used = a[:, :, 3].max(axis=1) > 0
a = a[used]
t_base = a[0, :, 0].min()
print('levels stamped: %d' % len(a))
print('level | per wave: fetch / work / wait-at-barrier (cycles) | level total')
for l in range(min(n, len(a))):
    cols = []
    for w in range(4):
        t0, t1, t2, t3 = a[l, w]
        cols.append('%5d/%5d/%5d' % (t1 - t0, t2 - t1, t3 - t2))
    print('%5d | %s | %6d' % (l, '  '.join(cols), a[l, :, 3].max() - a[l, :, 0].min()))
if len(a) > 8:
    tot = a[-1, :, 3].max() - a[0, :, 0].min()
    print('mean cycles per level over %d levels: %.0f' % (len(a), tot / len(a)))
    work = (a[:, :, 2] - a[:, :, 1]).max(axis=1)
    fetch = (a[:, :, 1] - a[:, :, 0]).max(axis=1)
    print('mean of the slowest wave per level: fetch %.0f, work %.0f cycles' % (fetch.mean(), work.mean()))
