#!/bin/bash
# SQ counters of the R1CS row kernel on the C5 bench (own run, kernel trace only): tools/pmc_c5.sh <out dir>
OUT=${1:-gpurun_out/pmc_c5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $ROOT/$OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_c5_sq
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d /tmp/pmc_c5_sq -- python3 $ROOT/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/pmc_c5_sq.log 2>&1
python3 $ROOT/tools/pmc_summary.py $(find /tmp/pmc_c5_sq -name '*counter_collection.csv' | head -1) 'r1cs_row_kernel<8, false, false>' 1000 > $ROOT/$OUT/pmc_c5_sq_counters.json
cat $ROOT/$OUT/pmc_c5_sq_counters.json
