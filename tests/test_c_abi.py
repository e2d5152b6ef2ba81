"""The public header is plain C and a C program can drive the recording half without a GPU."""
import os
import subprocess

import pytest

from helpers import REF_EXAMPLES, ROOT
import zkinterface_ir_amd as zk


def test_header_compiles_as_c99_and_c_example_records(tmp_path):
    exe = str(tmp_path / 'evaluate_workspace')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-pedantic', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'evaluate_workspace.c'),
                           '-L', os.path.dirname(zk.LIB_PATH), '-lzkgpu',
                           '-Wl,-rpath,' + os.path.dirname(zk.LIB_PATH), '-o', exe])
    out = subprocess.check_output([exe, '--record-only'] + REF_EXAMPLES, text=True)
    assert 'backend calls 210 (asserts 2)' in out


@pytest.mark.gpu
def test_c_example_and_cli_print_the_reference_verdicts(tmp_path):
    import io
    from zkinterface_ir_amd import cli
    exe = str(tmp_path / 'evaluate_workspace')
    subprocess.check_call(['gcc', '-std=c99', '-I', os.path.join(ROOT, 'include'),
                           os.path.join(ROOT, 'examples', 'evaluate_workspace.c'),
                           '-L', os.path.dirname(zk.LIB_PATH), '-lzkgpu',
                           '-Wl,-rpath,' + os.path.dirname(zk.LIB_PATH), '-o', exe])
    r = subprocess.run([exe] + REF_EXAMPLES, capture_output=True, text=True)
    assert r.returncode == 0 and 'The statement is TRUE!' in r.stderr
    r = subprocess.run([exe, '--valid-eval-metrics'] + REF_EXAMPLES, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert 'The statement is COMPLIANT with the specification!' in r.stderr and 'The statement is TRUE!' in r.stderr
    import json
    assert json.loads(r.stdout[r.stdout.index('{'):])['gate_stats']['functions_called'] == 4
    err = io.StringIO()
    assert cli.main(['evaluate'] + REF_EXAMPLES, err=err) == 0
    assert err.getvalue() == '\nThe statement is TRUE!\n'
    # the incorrect statement: cli.rs:557-571 text and a non-zero exit
    import circuits
    d = tmp_path / 'bad'
    d.mkdir()
    for name, buf in zip(('000_instance.sieve', '001_witness.sieve', '002_relation.sieve'),
                         circuits.golden_case('arith_101_incorrect')):
        (d / name).write_bytes(buf)
    err = io.StringIO()
    assert cli.main(['evaluate', str(d)], err=err) == 1
    assert err.getvalue() == ('\nThe statement is NOT TRUE!\nViolations:\n'
                              '- Wire_9 (may be weighted) should be 0, while it is not\n\n'
                              'Error: Found 1 violations.\n')
    r = subprocess.run([exe, str(d)], capture_output=True, text=True)
    assert r.returncode == 1 and 'Wire_9 (may be weighted) should be 0, while it is not' in r.stderr


def test_evaluator_template_runs_as_verifier_with_a_stub_backend(tmp_path):
    """evaluator.rs:1007-1080: arbitrary backend (Wire = i64, all zeros), no witness message."""
    import circuits
    csrc = os.path.join(ROOT, 'zkinterface-ir_amd', 'csrc')
    exe = str(tmp_path / 'verifier')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-I', csrc, os.path.join(ROOT, 'tests', 'cpp', 'test_evaluator_verifier.cpp'),
                           os.path.join(csrc, 'sieve', 'reader.cpp'), '-o', exe])
    inst, wit, rel = circuits.golden_case('arith_101_correct')
    (tmp_path / 'i.sieve').write_bytes(inst)
    (tmp_path / 'r.sieve').write_bytes(rel)
    out = subprocess.check_output([exe, str(tmp_path / 'i.sieve'), str(tmp_path / 'r.sieve')], text=True)
    # same backend-call count as the plaintext run (277 value calls, 6 asserts); all 6 witnesses absent
    assert out.strip() == 'calls 277 asserts 6 witnesses_without_value 6 violations 0'


def test_library_is_built_from_the_sources_in_the_tree():
    """lib/libzkgpu.so is git-ignored but travels to the GPU box: __graft_entry__.build() records a digest of the
    sources it compiled, smoke() / bench.py rebuild when it differs (ensure_built), and no tier tests a stale library"""
    import __graft_entry__ as entry
    assert entry.library_is_current(), 'lib/libzkgpu.so was not built from these sources: run __graft_entry__.build()'
    subprocess.check_call(['make', '-q', '-C', entry.PKG_DIR])      # nothing left for make to do


@pytest.mark.gpu
def test_library_on_the_gpu_box_is_built_from_the_sources_in_the_tree():
    import __graft_entry__ as entry
    assert entry.library_is_current()


def test_helper_thread_placement_unit(tmp_path):
    """tests/cpp/test_affinity.cpp: the CPUs that share a last-level cache with a CPU (csrc/affinity.hpp) hold that CPU and
    nothing outside the process's mask; a thread that follows another ends up allowed next to it; the caller's own mask is
    never touched; with ZKI_THREAD_AFFINITY=0 nothing is placed"""
    import subprocess
    csrc = os.path.join(ROOT, 'zkinterface-ir_amd', 'csrc')
    exe = str(tmp_path / 'test_affinity')
    subprocess.check_call(['g++', '-std=c++17', '-O2', '-pthread', '-I', csrc, os.path.join(ROOT, 'tests', 'cpp', 'test_affinity.cpp'), '-o', exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and 'bad=0' in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ZKI_THREAD_AFFINITY='0'))
    assert r.returncode == 0 and 'not readable here' in r.stdout, r.stdout + r.stderr
