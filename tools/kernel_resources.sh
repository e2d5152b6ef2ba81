#!/bin/bash
# Register / occupancy report of the gfx950 kernels of one field width (default 8 words = 256 bits):
#   tools/kernel_resources.sh [words]
W=${1:-8}
cd "$(dirname "$0")/../zkinterface-ir_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DZKGPU_W=$W -c -o /tmp/zk_kres_$W.o csrc/kernels_arith.hip \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = None
for line in sys.stdin:
    m = re.search(r"remark: .*?: (Function Name|VGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        m = re.search(r"(Function Name|    VGPRs|    SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = v; print()
        import subprocess
        print(subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0], end=": ")
    else:
        print("%s=%s" % (k.split(" ")[0], v), end=" ")
print()
'
