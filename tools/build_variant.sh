#!/bin/bash
# Build a variant of libzkgpu.so whose GF(2) kernels are compiled with extra -D flags (kernel experiments):
#   tools/build_variant.sh NAME "-D..."   ->  zkinterface-ir_amd/lib/variants/libzkgpu_NAME.so
# tools/c4_diag.py loads it with ZKGPU_VARIANT=NAME.  The product library (lib/libzkgpu.so) is not touched.
set -e
cd "$(dirname "$0")/../zkinterface-ir_amd"
name=$1; flags=$2
make -s -j8
mkdir -p obj/variants lib/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-inline-asm --offload-arch=gfx950 $flags -c -o obj/variants/kernels_bool_$name.o csrc/kernels_bool.hip
objs=$(ls obj/*.o obj/sieve/*.o | grep -v kernels_bool)
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o lib/variants/libzkgpu_$name.so $objs obj/variants/kernels_bool_$name.o
echo built lib/variants/libzkgpu_$name.so
