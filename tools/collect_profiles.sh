#!/bin/bash
# Collect the evidence `profiles/` holds for one round, on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r01'
# bench lines for c2 / c4 / c5, rocprofv3 kernel stats of the same commands, and the two --pmc passes
# (FETCH_SIZE, WRITE_SIZE; separate runs, kernel trace only) that tools/pmc_traffic.py turns into bytes per launch.
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for wl in c2 c4 c5; do
  extra=""
  [ $wl != c2 ] && extra="--workload $wl"
  echo "[collect] bench $wl"
  timeout -k 10 500 python3 $ROOT/bench.py --steps 20 --warmup 3 $extra > $OUT/bench_$wl.json 2> $OUT/bench_$wl.err
  rm -rf /tmp/prof_$wl
  echo "[collect] rocprofv3 stats $wl"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline $extra > $OUT/prof_$wl.log 2>&1
  cp $(find /tmp/prof_$wl -name '*kernel_stats.csv' | head -1) $OUT/${wl}_kernel_stats.csv
done
echo "[collect] pmc passes (c2, one stream)"
rm -rf /tmp/pmc_f /tmp/pmc_w
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/pmc_f.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --streams 1 > $OUT/pmc_w.log 2>&1
F=$(find /tmp/pmc_f -name '*counter_collection.csv' | head -1)
W=$(find /tmp/pmc_w -name '*counter_collection.csv' | head -1)
python3 $ROOT/tools/pmc_traffic.py $F $W $OUT/pmc_traffic.json 1000000 'replay_fused_kernel<8' c2 > /dev/null
for wl in c4 c5; do
  echo "[collect] pmc passes ($wl)"
  rm -rf /tmp/pmc_f_$wl /tmp/pmc_w_$wl
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$wl -- python3 $ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_f_$wl.log 2>&1
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$wl -- python3 $ROOT/bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_w_$wl.log 2>&1
  K='bool_lds_kernel'; [ $wl = c5 ] && K='r1cs_row_kernel<8, false>'
  python3 $ROOT/tools/pmc_traffic.py $(find /tmp/pmc_f_$wl -name '*counter_collection.csv' | head -1) $(find /tmp/pmc_w_$wl -name '*counter_collection.csv' | head -1) $OUT/pmc_traffic_$wl.json 1000 "$K" $wl > /dev/null
done
head -4 $F > $OUT/pmc_fetch_sample.csv
head -4 $W > $OUT/pmc_write_sample.csv
echo "[collect] done"
tail -c 600 $OUT/bench_c2.json
