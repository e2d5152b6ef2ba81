#!/bin/bash
# C2 launch-geometry sweep on one box: streams x entries per wave (ms per step of the timed steps only)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { python3 $ROOT/bench.py --timed-steps-only --steps 20 --warmup 3 "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('ms_per_step %.3f' % d['ms_per_step'], 'kernel_ms %.3f' % (d['roofline']['avg_launch_ms'] * d['roofline']['launches_per_step']))"; }
for s in 1 2 3 4; do echo "== --streams $s"; run --streams $s; done
for w in 2 3; do echo "== ZKI_OPW=$w --streams 2"; ZKI_OPW=$w run --streams 2; done
echo "== --streams 2 --lane-group 512"; run --streams 2 --lane-group 512
echo "== --streams 4 --lane-group 512"; run --streams 4 --lane-group 512
echo "== ZKI_XCD_MAP=0 --streams 2"; ZKI_XCD_MAP=0 run --streams 2
