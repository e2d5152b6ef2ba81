#!/bin/bash
# A/B of two builds of libzkgpu.so on ONE box, back to back (box-to-box variation is as large as most kernel changes):
#   tools/ab_bench.sh build/ab/libzkgpu_head.so [bench args...]        (default: --timed-steps-only --steps 20 --warmup 3)
# prints ms_per_step of B (the library in the tree), A (the other file), B, A.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
A=$ROOT/$1; shift
ARGS=${@:---timed-steps-only --steps 20 --warmup 3}
LIB=$ROOT/zkinterface-ir_amd/lib/libzkgpu.so
cp $LIB /tmp/ab_B.so
run() { python3 $ROOT/bench.py $ARGS 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', 'ms_per_step %.3f' % d['ms_per_step'], 'kernel_ms %.3f' % (d['roofline']['avg_launch_ms'] * d['roofline']['launches_per_step']), 'value %.4g' % d['value'])"; }
for round in 1 2; do
  cp /tmp/ab_B.so $LIB; run B
  cp $A $LIB; run A
done
cp /tmp/ab_B.so $LIB
