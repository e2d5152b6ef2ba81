"""Guards against toolchain drift (CPU tier: hipcc cross-compiles without a GPU).

Two kernels rest on what hipcc allocates, not only on what the C++ says:
  * `bool_lds_kernel<BR>` (device/bool_kernels.hpp) is compiled for 64 VGPRs and keeps everything that is in flight in
    v64..v125, registers named in its asm text; the kernel descriptor only covers them because the empty
    `asm volatile("" ::: "v125")` clobber makes hipcc allocate up to v125.  With launch_bounds(1024) a wave owns exactly
    128 unified registers: anything else (fewer allocated, AGPRs, scratch) means waves touching registers that are not
    theirs, i.e. silently wrong GF(2) verdicts.
  * `replay_fused_kernel<8, kFusedHot>` (the C2 headline kernel) must stay at <= 64 VGPRs (8 waves per SIMD) without
    scratch; the row kernel must not spill.
The same numbers are printed by tools/kernel_resources.py."""
import os
import re
import sys

import pytest

from helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, 'tools'))
import kernel_resources  # noqa: E402

BOOL_HPP = os.path.join(ROOT, 'zkinterface-ir_amd', 'csrc', 'device', 'bool_kernels.hpp')


def _highest_hand_managed_register():
    """kRegH + 1 of bool_kernels.hpp: the highest register its asm text names"""
    text = open(BOOL_HPP).read()
    m = re.search(r'kRegH\s*=\s*(\d+)', text)
    assert m, 'bool_kernels.hpp no longer defines kRegH'
    top = int(m.group(1)) + 1
    clobber = re.search(r'asm volatile\(""\s*:::\s*"v(\d+)"\)', text)
    assert clobber and int(clobber.group(1)) == top, 'the allocation clobber must name the highest hand-managed register'
    compiler = int(re.search(r'kLdsCompilerVgprs\s*=\s*(\d+)', text).group(1))
    first = int(re.search(r'kRegP\s*=\s*(\d+)', text).group(1))
    assert first >= compiler, 'hand-managed registers start inside the range hipcc allocates from'
    return top


@pytest.fixture(scope='module')
def bool_kernels():
    return kernel_resources.resources('kernels_bool.hip')


def _check_lds_kernels(res, top):
    lds = {n: k for n, k in res.items() if 'bool_lds_kernel<' in n}
    assert lds, 'no bool_lds_kernel instantiation found in the remarks'
    for name, k in lds.items():
        assert k['vgprs'] >= top + 1, (name, 'allocated VGPRs do not cover v%d' % top, k)
        assert k['agprs'] == 0, (name, k)
        assert k['vgprs'] + k['agprs'] <= 128, (name, 'more than the 128 unified registers a 1024-thread workgroup gets', k)
        assert k['scratch'] == 0 and k['vgpr_spill'] == 0 and k['sgpr_spill'] == 0, (name, k)
        assert k['occupancy'] == 4, (name, k)     # 16 waves of one workgroup on 4 SIMDs
    return lds


def test_bool_lds_kernel_owns_the_registers_its_asm_names(bool_kernels):
    lds = _check_lds_kernels(bool_kernels, _highest_hand_managed_register())
    # every block size the engine may pick (kernels_bool.hip ZKGPU_LDS_BLOCK_ROWS)
    assert sorted(int(re.search(r'<(\d+)>', n).group(1)) for n in lds) == [4, 6, 8, 9, 10, 12]


def test_other_boolean_kernels_do_not_spill(bool_kernels):
    for name, k in bool_kernels.items():
        assert k['scratch'] == 0 and k['vgpr_spill'] == 0, (name, k)


def test_arithmetic_kernels_at_bn254_width():
    res = kernel_resources.resources('kernels_arith.hip', ['-DZKGPU_W=8'])
    hot = res['zkgpu::replay_fused_kernel<8, 0>']
    assert hot['vgprs'] <= 64 and hot['agprs'] == 0 and hot['scratch'] == 0 and hot['occupancy'] == 8 and hot['sgpr_spill'] == 0, hot
    row = res['zkgpu::r1cs_row_kernel<8, false, false>']
    assert row['scratch'] == 0 and row['vgpr_spill'] == 0 and row['sgpr_spill'] == 0, row
    # (the instantiation with the coefficient classes is a separate one: the BASELINE rows keep the registers of their path)
    assert res['zkgpu::r1cs_row_kernel<8, false, true>']['scratch'] == 0
    for name, k in res.items():
        # (the input streams' descriptors sit behind one pointer, device/args.hpp InputAux: in the kernarg block they made
        # the cold kernels spill 28 SGPRs and the strands of a structured relation 13 % slower)
        # (the strand kernel keeps a program entry in SGPRs while another one runs: hipcc parks three wave-uniform conditions
        # there -- lane < batch and the two forms of "the input buffers hold N-word values" --, six SGPRs in the lanes of one
        # VGPR, written once per launch and read back only by the entries that load an input, never by the Add/Mul entries of
        # the dependency chain; anything beyond that is a regression)
        allowed = 6 if 'replay_strand_kernel' in name else 0
        assert k['scratch'] == 0 and k['vgpr_spill'] == 0 and k['sgpr_spill'] <= allowed, (name, k)


def test_any_modulus_kernels_private_memory():
    """the operands of the any-modulus kernels are thread-private arrays indexed at run time (device/generic_kernels.hpp):
    one instantiation per capacity class, so that a 600-bit modulus does not pay for a 4096-bit one -- registers for the
    small classes, scratch memory bounded by the class for the large ones"""
    res = kernel_resources.resources('kernels_generic.hip')
    caps = {}
    small = {}
    for name, k in res.items():
        m = re.search(r'replay_generic_kernel<(\d+), (\d+)>', name)
        if m and int(m.group(2)) == 0:
            caps[int(m.group(1))] = k
        elif m:
            small[int(m.group(2))] = k
    assert sorted(caps) == [16, 32, 64, 128]
    # characteristics of up to eight words: word counts at compile time, everything in registers (SmallParams)
    assert sorted(small) == [1, 2, 3, 4, 5, 6, 7, 8]
    for kc, k in small.items():
        assert k['scratch'] == 0 and k['agprs'] == 0 and k['occupancy'] >= 3, (kc, k)
    for cap, k in caps.items():
        assert k['vgprs'] + k['agprs'] <= 512 and k['occupancy'] >= 1, (cap, k)
        assert k['scratch'] <= 40 * cap, (cap, k)     # 3.6 KB per lane at 4096 bits
