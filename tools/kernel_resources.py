#!/usr/bin/env python3
"""Registers, scratch and occupancy of the gfx950 kernels as hipcc allocates them (no GPU needed):

  tools/kernel_resources.py [words ...]       report for kernels_arith.hip at those widths (default 8) + kernels_bool.hip
  tools/kernel_resources.py --flags "-D..."   extra flags for kernels_bool.hip (the tools/build_variant.sh experiments)

`resources(source, flags)` compiles one translation unit for the device only with
-Rpass-analysis=kernel-resource-usage and returns {demangled kernel name: {vgprs, agprs, sgprs, scratch, occupancy,
lds, vgpr_spill, sgpr_spill}}.  tests/test_kernel_resources.py (CPU tier) asserts on it the properties the hand-scheduled
kernels rest on, so that a toolchain change shows up as a failing test and not as wrong verdicts on the GPU."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'zkinterface-ir_amd')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# the flags of zkinterface-ir_amd/Makefile (kernels_bool.hip adds -Wno-inline-asm there too)
BASE_FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '--cuda-device-only', '-Wno-inline-asm']

_FIELDS = {'VGPRs': 'vgprs', 'AGPRs': 'agprs', 'TotalSGPRs': 'sgprs', 'ScratchSize [bytes/lane]': 'scratch',
           'Occupancy [waves/SIMD]': 'occupancy', 'LDS Size [bytes/block]': 'lds', 'VGPRs Spill': 'vgpr_spill',
           'SGPRs Spill': 'sgpr_spill'}


def _demangle(names):
    out = subprocess.run(['c++filt'] + names, capture_output=True, text=True, check=True).stdout.split('\n')
    return [re.sub(r'^void ', '', o).split('(')[0] for o in out[:len(names)]]


def resources(source, flags=()):
    src = source if os.path.isabs(source) else os.path.join(PKG, 'csrc', source)
    cmd = [HIPCC] + BASE_FLAGS + list(flags) + ['-c', '-o', os.devnull, src, '-Rpass-analysis=kernel-resource-usage']
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=PKG)
    if p.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + p.stderr[-4000:])
    kernels, cur = [], None
    for line in p.stderr.split('\n'):
        m = re.search(r'remark:\s+([A-Za-z][^:]*?): (\S+) \[-Rpass-analysis', line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == 'Function Name':
            cur = {'mangled': val}
            kernels.append(cur)
        elif cur is not None and key in _FIELDS:
            cur[_FIELDS[key]] = int(val)
    names = _demangle([k['mangled'] for k in kernels]) if kernels else []
    return {n: k for n, k in zip(names, kernels)}


def main():
    args = sys.argv[1:]
    flags = []
    if '--flags' in args:
        i = args.index('--flags')
        flags = args[i + 1].split()
        del args[i:i + 2]
    widths = [int(a) for a in args] or [8]
    for w in widths:
        print('== kernels_arith.hip -DZKGPU_W=%d' % w)
        for name, k in resources('kernels_arith.hip', ['-DZKGPU_W=%d' % w]).items():
            print('  %-58s VGPRs=%-3d AGPRs=%-2d SGPRs=%-3d scratch=%-3d occupancy=%d' %
                  (name, k['vgprs'], k['agprs'], k['sgprs'], k['scratch'], k['occupancy']))
    print('== kernels_bool.hip %s' % ' '.join(flags))
    for name, k in resources('kernels_bool.hip', flags).items():
        print('  %-58s VGPRs=%-3d AGPRs=%-2d SGPRs=%-3d scratch=%-3d occupancy=%d LDS=%d' %
              (name, k['vgprs'], k['agprs'], k['sgprs'], k['scratch'], k['occupancy'], k['lds']))
    print('== kernels_generic.hip')
    for name, k in resources('kernels_generic.hip').items():
        print('  %-58s VGPRs=%-3d AGPRs=%-2d SGPRs=%-3d scratch=%-4d occupancy=%d' %
              (name, k['vgprs'], k['agprs'], k['sgprs'], k['scratch'], k['occupancy']))


if __name__ == '__main__':
    main()
