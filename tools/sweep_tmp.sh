for opw in 1 2 3 4; do
  echo "== interleaved opw=$opw"
  ZKI_OPW=$opw timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value']/1e9)"
done
