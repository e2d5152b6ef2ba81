"""CPU checkers for the full-size BASELINE workloads (test infrastructure, like oracle/).

They evaluate the *definition* of a synthetic workload (zkinterface_ir_amd/workloads.py: which gate reads which
wire), not the recorded tape and not the device program, so they share nothing with the product but the workload
generator.  Each is pinned to the oracle (oracle/zki_oracle.cpp, the restatement of
rust/src/consumers/evaluator.rs) in the CPU tier on small instances and, at full size, through the committed
digests of tests/golden/ (make_c4_digests.py, make_c5_digests.py).
"""
import hashlib

import numpy as np


def bool_layered_outputs(wl, inst, wit):
    """BoolLayered (C4) for a whole batch, 64 witnesses per uint64 word, one numpy gather per layer.
    and / xor on bits; `not` = is_zero ? 1 : 0 (evaluator.rs:924-938), i.e. the complement of a {0,1} value.
    Returns the n_out output bits per lane: uint8 [batch][n_out]."""
    batch = inst.shape[0]
    words = (batch + 63) // 64
    bits = np.zeros((wl.W, words * 64), dtype=np.uint8)
    bits[:wl.n_instance0, :batch] = inst[:, :wl.n_instance0, 0].T
    bits[wl.n_instance0:, :batch] = wit[:, :, 0].T
    cur = np.packbits(bits, axis=1, bitorder='little').view(np.uint64)       # [W][words]
    for l in range(wl.D):
        a = cur[wl.src_a[l]]
        b = cur[wl.src_b[l]]
        k = wl.kind[l][:, None]
        cur = np.where(k == 8, a & b, np.where(k == 9, a ^ b, ~a))
    out = np.unpackbits(cur[:wl.n_out].view(np.uint8), axis=1, bitorder='little')[:, :batch]
    return np.ascontiguousarray(out.T)


def bool_digest(bits_row):
    """digest of one lane's output bits (the form tests/golden/c4_digests.json stores)"""
    return hashlib.sha256(''.join(str(int(b)) for b in bits_row).encode()).hexdigest()


def r1cs_lane_assignment(wl, w_lane):
    """R1csSynthetic (C5), one lane, Python integers: z_r = <a,w> * <b,w> mod p for every product row r, by the
    mathematical definition.  Returns the list of all n_base + 1 + M variable values (E left as given)."""
    p = wl.p
    coefs = [int.from_bytes(wl.coefs[i].tobytes(), 'little') for i in range(len(wl.coefs))]
    val = [int.from_bytes(w_lane[k].tobytes(), 'little') for k in range(wl.n_witness)] + [0] * wl.M
    picks = wl.picks.tolist()
    cidx = wl.coef_idx.tolist()
    base = wl.n_base + 1
    for r in range(wl.M):
        v, c = picks[r], cidx[r]
        a = coefs[c[0]] * val[v[0]] + coefs[c[1]] * val[v[1]] + coefs[c[2]] * val[v[2]]
        b = coefs[c[3]] * val[v[3]] + coefs[c[4]] * val[v[4]] + coefs[c[5]] * val[v[5]]
        val[base + r] = (a % p) * (b % p) % p
    return val


def r1cs_sample_vars(wl, n=257):
    """the variables whose values a C5 digest covers: n product-row variables spread over all levels + the last z"""
    step = max(1, wl.M // (n - 1))
    ids = [wl.n_base + 1 + r for r in range(0, wl.M, step)][:n - 1]
    return ids + [wl.last_z]


def r1cs_digest(values, width):
    return hashlib.sha256(b''.join(int(v).to_bytes(width, 'little') for v in values)).hexdigest()
