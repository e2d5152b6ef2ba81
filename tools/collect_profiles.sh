#!/bin/bash
# Collect the evidence `profiles/` holds for one round, on the GPU box, in four calls (each under gpurun's limit):
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r02 bench'   bench lines + rocprofv3 kernel stats
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r02 pmc_c2'  --pmc passes of the C2 kernel (+ the 4096-lane variant)
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r02 pmc_c45' --pmc passes of the GF(2) and R1CS kernels
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r02 pmc_c5s' --pmc passes of the R1CS kernel on small coefficients
# Counters are collected in runs of their own with --kernel-trace only (FETCH_SIZE and WRITE_SIZE never share a pass);
# tools/pmc_traffic.py applies the gfx950 corrections of MI355X_MICROARCH.md, tools/binding_evidence.py writes
# profiles/binding_<workload>.json from the summaries.  Copy gpurun_out/<tag>/* into profiles/ afterwards.
set -e
TAG=${1:-r04}
WHAT=${2:-bench}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python3 $ROOT/bench.py"
pmc() {  # pmc <name> <counters...> -- <bench args...>
  local name=$1; shift
  local counters=()
  while [ "$1" != "--" ]; do counters+=("$1"); shift; done
  shift
  rm -rf /tmp/pmc_$name
  echo "[collect] pmc $name: ${counters[*]}"
  timeout -k 10 420 rocprofv3 --pmc "${counters[@]}" --kernel-trace --output-format csv -d /tmp/pmc_$name -- $B "$@" > $OUT/pmc_$name.log 2>&1
  cp $(find /tmp/pmc_$name -name '*counter_collection.csv' | head -1) /tmp/pmc_$name.csv
}
if [ $WHAT = bench ]; then
  # the driver's command: the C2 headline line with the c4 / c5 / structured lines under `secondary`
  echo "[collect] bench (default invocation)"
  # (the line is the compact one the driver records; everything else is in the detail file the run writes)
  ZKI_BENCH_DETAIL=gpurun_out/$TAG/${TAG}_bench_detail.json timeout -k 10 900 $B --steps 20 --warmup 3 > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
  python3 - $OUT $TAG <<'PY'
import json, sys
out, tag = sys.argv[1], sys.argv[2]
line = open('%s/%s_bench_default.json' % (out, tag)).read()
print('[collect] bench line: %d characters' % len(line))
d = json.load(open('%s/%s_bench_detail.json' % (out, tag)))
sec = d.pop('secondary', {})
json.dump(d, open('%s/%s_bench_c2.json' % (out, tag), 'w'))
for k, v in sec.items():
    json.dump(v, open('%s/%s_bench_%s.json' % (out, tag, k), 'w'))
PY
  # kernel stats of the TIMED REGION only: --timed-steps-only runs nothing but the probe (another kernel: the unfused
  # replay_kernel / the HBM-table GF(2) kernel), the warm-up steps and the timed steps
  for wl in c2 c4 c5; do
    extra=""; [ $wl != c2 ] && extra="--workload $wl"
    rm -rf /tmp/prof_$wl
    echo "[collect] rocprofv3 stats $wl"
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$wl -- $B --steps 20 --warmup 3 --timed-steps-only $extra > $OUT/${TAG}_bench_${wl}_under_trace.json 2> $OUT/prof_$wl.log
    cp $(find /tmp/prof_$wl -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_${wl}_kernel_stats.csv
  done
  tail -c 400 $OUT/${TAG}_bench_c2.json
elif [ $WHAT = pmc_c2 ]; then
  # one stream: a level is ONE launch over all 1024 witnesses (grid > 2M threads); the probe session of bench.py replays
  # its lane halves on two streams (1.3M threads per launch) and is left out by the grid filter
  C2="--steps 3 --warmup 1 --timed-steps-only --streams 1"
  K='replay_fused_kernel<8, 0>'
  pmc c2_f FETCH_SIZE -- $C2
  pmc c2_w WRITE_SIZE -- $C2
  python3 $ROOT/tools/pmc_traffic.py /tmp/pmc_c2_f.csv /tmp/pmc_c2_w.csv $OUT/pmc_traffic_latest.json 2000000 "$K" c2 > /dev/null
  head -4 /tmp/pmc_c2_f.csv > $OUT/${TAG}_pmc_fetch_sample.csv
  head -4 /tmp/pmc_c2_w.csv > $OUT/${TAG}_pmc_write_sample.csv
  pmc c2_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -- $C2
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_c2_sq.csv "$K" 2000000 > $OUT/${TAG}_pmc_c2_sq_counters.json
  pmc c2_tcc TCC_HIT_sum TCC_MISS_sum -- $C2
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_c2_tcc.csv "$K" 2000000 > $OUT/${TAG}_pmc_c2_tcc_counters.json
  # the same program with 4096 witnesses in flight (1.05 GB wire table: cannot sit in the Infinity Cache)
  H="--steps 2 --warmup 1 --timed-steps-only --streams 1 --batch-per-gpu 4096 --lane-group 4096"
  pmc c2h_f FETCH_SIZE -- $H
  pmc c2h_w WRITE_SIZE -- $H
  python3 $ROOT/tools/pmc_traffic.py /tmp/pmc_c2h_f.csv /tmp/pmc_c2h_w.csv $OUT/pmc_traffic_c2_hbm_variant.json 4000000 "$K" c2_hbm_variant > /dev/null
elif [ $WHAT = pmc_c4 ]; then   # (the LDS / SQ passes of the GF(2) kernel alone: kernel experiments)
  C4="--workload c4 --steps 2 --warmup 1 --timed-steps-only"
  LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU"
  pmc c4_lds $LDS -- $C4
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_c4_lds.csv bool_lds_kernel 1000 > $OUT/${TAG}_pmc_c4_lds_counters.json
  pmc c4_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -- $C4
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_c4_sq.csv bool_lds_kernel 1000 > $OUT/${TAG}_pmc_c4_sq_counters.json
elif [ $WHAT = pmc_c45 ]; then
  C4="--workload c4 --steps 2 --warmup 1 --timed-steps-only"
  pmc c4_f FETCH_SIZE -- $C4
  pmc c4_w WRITE_SIZE -- $C4
  python3 $ROOT/tools/pmc_traffic.py /tmp/pmc_c4_f.csv /tmp/pmc_c4_w.csv $OUT/pmc_traffic_c4.json 1000 bool_lds_kernel c4 > /dev/null
  LDS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU"
  pmc c4_lds $LDS -- $C4
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_c4_lds.csv bool_lds_kernel 1000 > $OUT/${TAG}_pmc_c4_lds_counters.json
  pmc c4_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY -- $C4
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_c4_sq.csv bool_lds_kernel 1000 > $OUT/${TAG}_pmc_c4_sq_counters.json
  export ZKI_BANK_AWARE=0   # the slot numbering of round 1, for the before / after of the bank-aware schedule
  pmc c4_lds_unbanked $LDS -- $C4
  unset ZKI_BANK_AWARE
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_c4_lds_unbanked.csv bool_lds_kernel 1000 > $OUT/${TAG}_pmc_c4_lds_counters_bank_unaware.json
  # the same relation shape with gate j reading wires j and j+1 of the previous layer: conflict-free by construction,
  # same program size -- what is left is the program stream and the barriers
  ZKI_C4_WIRING=identity ZKI_BENCH_DETAIL=gpurun_out/$TAG/${TAG}_bench_c4_identity_detail.json $B --workload c4 --steps 10 --warmup 2 --no-cpu-baseline --no-first-verdict > $OUT/${TAG}_bench_c4_identity_wiring.json 2> $OUT/bench_c4_identity.err
  C5="--workload c5 --steps 2 --warmup 1 --timed-steps-only"
  K5='r1cs_row_kernel<8, false, false>'
  pmc c5_f FETCH_SIZE -- $C5
  pmc c5_w WRITE_SIZE -- $C5
  python3 $ROOT/tools/pmc_traffic.py /tmp/pmc_c5_f.csv /tmp/pmc_c5_w.csv $OUT/pmc_traffic_c5.json 1000 "$K5" c5 > /dev/null
  pmc c5_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -- $C5
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_c5_sq.csv "$K5" 1000 > $OUT/${TAG}_pmc_c5_sq_counters.json
elif [ $WHAT = pmc_c5s ]; then
  # the rows of C5 with small coefficients (bench.py --coefs small): the instantiation with the coefficient classes
  C5S="--workload c5 --coefs small --steps 2 --warmup 1 --timed-steps-only"
  K5S='r1cs_row_kernel<8, false, true>'
  pmc c5s_f FETCH_SIZE -- $C5S
  pmc c5s_w WRITE_SIZE -- $C5S
  python3 $ROOT/tools/pmc_traffic.py /tmp/pmc_c5s_f.csv /tmp/pmc_c5s_w.csv $OUT/pmc_traffic_c5_small.json 1000 "$K5S" c5_small > /dev/null
  pmc c5s_sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -- $C5S
  python3 $ROOT/tools/pmc_summary.py /tmp/pmc_c5s_sq.csv "$K5S" 1000 > $OUT/${TAG}_pmc_c5_small_sq_counters.json
fi
echo "[collect] $WHAT done"
ls $OUT
