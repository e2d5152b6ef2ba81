"""CPU tier: IR -> R1CS emission of the product against the test-side restatement of
to_r1cs.rs, and satisfaction of the emitted system by the oracle's wire values."""
import pytest

import circuits
from helpers import golden_buffers
from oracle_lib import OracleRun
import r1cs_ref
import zkinterface_ir_amd as zk

CASES = [('ref_examples', 101), ('arith_101_correct', 101), ('arith_101_incorrect', 101),
         ('arith_bn254_correct', circuits.BN254_R)]


@pytest.mark.parametrize('use_correction', [False, True])
@pytest.mark.parametrize('name,p', CASES)
def test_rows_follow_the_converter_rules(name, p, use_correction):
    bufs = golden_buffers(name)
    ev = zk.Evaluator.from_messages(bufs)
    ev.r1cs_from_tape(use_correction)
    rows, var_of_op = ev.r1cs_export()
    kinds, a, b = ev.tape()
    ref_rows, ref_var_of = r1cs_ref.rows_from_tape(kinds, a, b, ev.constants(), p, use_correction)
    assert rows == ref_rows
    assert [None if int(v) == 2 ** 64 - 1 else int(v) for v in var_of_op] == ref_var_of
    info = ev.r1cs_info()
    n_ops = sum(1 for k in kinds if int(k) in (1, 2, 3, 4, 9, 10, 11, 12))
    assert info['rows'] == n_ops  # one BilinearConstraint per add/mul/addc/mulc/assert call (to_r1cs.rs)
    # the witness the converter would emit = the oracle's values; every row holds (exactly, with the
    # quotient "correction" wires; modulo p without them), except the row of a failing assert
    ref = OracleRun(buffers=bufs)
    vals = ref.trace_values()
    w = {0: 1}
    value_ops = [i for i, k in enumerate(kinds) if int(k) != 9]
    for t, i in enumerate(value_ops[:len(vals)]):
        w[ref_var_of[i]] = vals[t]
    if use_correction:
        for i in value_ops[:len(vals)]:  # quotient wires (to_r1cs.rs:183-185,235-237)
            k = int(kinds[i])
            if k in (1, 2, 3, 4):
                x = w[ref_var_of[int(a[i])]]
                y = w[ref_var_of[int(b[i])]] if k in (1, 2) else int.from_bytes(ev.constants()[int(b[i])], 'little')
                w[ref_var_of[i] + 1] = ((x + y) if k in (1, 3) else (x * y)) // p
    bad = []
    for r, (A, B, C) in enumerate(rows):
        if not all(v in w for v, _ in A + B + C):
            break  # the oracle stopped at the first failing assert
        lhs = r1cs_ref.lincomb(A, w) * r1cs_ref.lincomb(B, w)
        rhs = r1cs_ref.lincomb(C, w)
        if (lhs != rhs) if use_correction else ((lhs - rhs) % p != 0):
            bad.append(r)
    if ref.violations:
        assert len(bad) == 1 and rows[bad[0]][2] == [(0, 0)]  # exactly the assert row
    else:
        assert bad == []


def test_boolean_relation_is_refused():
    ev = zk.Evaluator.from_messages(golden_buffers('bool_correct'))
    with pytest.raises(zk.ZkGpuError):
        ev.r1cs_from_tape()


@pytest.mark.parametrize('p', [101, 2 ** 61 - 1, circuits.BN254_R])
def test_cpu_row_check_against_python_integers(p):
    """oracle/cpu_opt.cpp zko_r1cs_check (the C5 cpu_baseline and second checker of the row kernel): product-row
    assignment and the check of every row against plain Python integer arithmetic."""
    import numpy as np
    from oracle_lib import r1cs_check
    from zkinterface_ir_amd import workloads
    wl = workloads.R1csSynthetic(M=300, n_base=24, n_coefs=50, seed=7, p=p)
    row_ptr, tv, tc, cb = wl.csr()
    batch = 6
    w = wl.witnesses(batch)
    coefs = [int.from_bytes(cb[i].tobytes(), 'little') for i in range(len(cb))]
    expect = []
    for lane in range(batch):
        val = {k: int.from_bytes(w[lane, k].tobytes(), 'little') for k in range(wl.n_witness)}
        for r in range(wl.M):
            terms = [(int(tv[7 * r + k]), coefs[int(tc[7 * r + k])]) for k in range(6)]
            a = sum(c * val[v] for v, c in terms[:3]) % p
            b = sum(c * val[v] for v, c in terms[3:]) % p
            val[wl.n_base + 1 + r] = a * b % p
        e = val[wl.last_z] if lane % 2 == 0 else (val[wl.last_z] + 1) % p   # odd lanes: the comparison row fails
        w[lane, wl.n_base] = np.frombuffer(e.to_bytes(wl.width, 'little'), dtype=np.uint8)
        expect.append(0xFFFFFFFF if lane % 2 == 0 else wl.M)
    ff, _ = r1cs_check(row_ptr, tv, tc, cb, wl.mod_le, w, wl.n_base + 1 + wl.M, wl.M, 2)
    assert ff.tolist() == expect
