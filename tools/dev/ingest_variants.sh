mkdir -p gpurun_out/r04
run() { echo "== $*"; env "$@" ONE_BUFFER=1 ZKI_SCHED_PROFILE=1 python tools/dev/c4_host_profile.py 2>&1 | grep "finalize\] \|threads" | cut -c1-100; }
(
for k in 1 2 3 4; do run THREADS=8 STREAM=1; done
for k in 1 2 3; do run THREADS=8; done
run THREADS=8 STREAM=1 ZKI_THREAD_AFFINITY=0
) > gpurun_out/r04/ingest_variants6.txt 2>&1
cat gpurun_out/r04/ingest_variants6.txt
for k in 1 2 3; do python bench.py --workload c4 --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('c4 first verdict', d['config']['first_verdict_s'])"; done
