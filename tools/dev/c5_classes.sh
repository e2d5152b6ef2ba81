mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "r1cs" > gpurun_out/r04/pytest_gpu_9.log 2>&1; echo pytest rc $?; tail -15 gpurun_out/r04/pytest_gpu_9.log
for c in random small; do python bench.py --workload c5 --coefs $c --no-cpu-baseline --no-first-verdict --steps 5 --warmup 1 2>gpurun_out/r04/c5_$c.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('c5 $c ms_per_step', d['ms_per_step'], d['config'].get('satisfied'), d['roofline']['frac_algorithmic'], d['config'].get('combinations_by_coefficient_class'))"; done
