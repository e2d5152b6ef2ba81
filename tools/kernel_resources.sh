#!/bin/bash
# Register / occupancy report of the gfx950 kernels (tools/kernel_resources.py does the work and is what
# tests/test_kernel_resources.py asserts on):   tools/kernel_resources.sh [words ...]
exec python3 "$(dirname "$0")/kernel_resources.py" "$@"
