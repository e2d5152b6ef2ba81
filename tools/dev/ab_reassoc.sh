mkdir -p gpurun_out/r04
timeout -k 10 600 python -m pytest tests/test_strands.py tests/test_gpu_parity.py tests/test_stream.py tests/test_field_segments.py tests/test_unreduced.py -m gpu -x -q > gpurun_out/r04/pytest_gpu_17.log 2>&1; echo pytest rc $?; tail -3 gpurun_out/r04/pytest_gpu_16.log
for ra in 1 0 1 0; do ZKI_STRAND_SPLIT=$ra python bench.py --workload structured --chained --no-cpu-baseline --no-first-verdict --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('chained split $ra ms_per_step', d['ms_per_step'], d['config']['satisfied'])"; done
ZKI_STRAND_REASSOC=1 python bench.py --workload structured --no-cpu-baseline --no-first-verdict --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('structured ms_per_step', d['ms_per_step'], d['config']['satisfied'])"
(cd zkinterface-ir_amd && touch csrc/kernels_arith.hip && make -s -j8 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DZKGPU_STRAND_STAMPS" 2>&1 | grep -v warning | head -5)
ZKGPU_STRAND_STAMPS=/tmp/stamps.bin python bench.py --workload structured --chained --no-cpu-baseline --no-first-verdict --steps 2 --warmup 1 > /dev/null 2>&1
python tools/strand_stamps.py /tmp/stamps.bin 24 > gpurun_out/r04/strand_stamps_split.txt 2>&1
sed -n 8,20p gpurun_out/r04/strand_stamps_split.txt
