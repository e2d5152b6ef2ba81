set -e
mkdir -p gpurun_out/r3
rm -f gpurun_out/r3/sweep4.txt
for cfg in "2 1" "1 1" "2 0" "0 1"; do
  set -- $cfg
  echo "== sort=$1 xcd_map=$2" >> gpurun_out/r3/sweep4.txt
  ZKI_SORT_BY_OPERAND=$1 ZKI_XCD_MAP=$2 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value']/1e9, d['config'].get('host_seconds'))" >> gpurun_out/r3/sweep4.txt
done
cat gpurun_out/r3/sweep4.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
