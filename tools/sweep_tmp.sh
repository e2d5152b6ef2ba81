set -e
mkdir -p gpurun_out/r3
rm -f gpurun_out/r3/sweep6.txt
for cfg in "--batch-per-gpu 1024" "--batch-per-gpu 4096" "--batch-per-gpu 1536" "--batch-per-gpu 1000" "--batch-per-gpu 512" "--batch-per-gpu 8192"; do
  echo "== $cfg" >> gpurun_out/r3/sweep6.txt
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $cfg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value']/1e9)" >> gpurun_out/r3/sweep6.txt
done
cat gpurun_out/r3/sweep6.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
