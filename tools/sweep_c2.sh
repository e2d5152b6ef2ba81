#!/bin/bash
# C2 tuning sweep on the GPU box: resident waves per SIMD of the Add/Mul kernel x entries per wave.
#   gpurun -- 'bash tools/sweep_c2.sh gpurun_out/r2b/sweep.txt'
OUT=${1:-gpurun_out/sweep_c2.txt}
mkdir -p $(dirname $OUT)
: > $OUT
for hw in 0 7 6 5 4; do
  for opw in 1 2 3; do
    r=$(ZKI_HOT_WAVES=$hw ZKI_OPW=$opw timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("%.3f ms  %.1f G/s" % (d["ms_per_step"], d["value"]/1e9))')
    echo "hot_waves=$hw opw=$opw  $r" | tee -a $OUT
  done
done
