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


# ---- every lane of the full-size workloads (tests/golden/c2_all_lanes.json, c4_all_lanes.json) -----------------------
def lane_hash(outputs_row):
    """16 hex digits for one lane: the first 8 bytes of the SHA-256 of its output values as little-endian bytes
    ([n_out][width] uint8 for GF(p); [n_out] bits, one byte each, for GF(2))"""
    return hashlib.sha256(np.ascontiguousarray(outputs_row, dtype=np.uint8).tobytes()).hexdigest()[:16]


def arith_layered_outputs(wl, inst, wit, threads=None):
    """ArithLayered (C2) for a whole batch: the n_out output wires of every lane, uint8 [batch][n_out][width], by
    `cpu_opt` (oracle/cpu_opt.cpp: flat array + 4x64 Montgomery, a different arithmetic from both the literal oracle's
    long division and the GPU's 32-bit product scanning) on the tape the host records for the relation WITHOUT its
    epilogue.  Every directive of that relation is one value-returning backend call, so tape index == wire id."""
    import os
    import oracle_lib
    import zkinterface_ir_amd as zk
    assert wl.width == 32
    ev = zk.Evaluator()
    ev.declare_inputs(wl.n_instance0, wl.n_witness)
    for m in wl.relation_messages(with_epilogue=False, free_last=False):
        ev.ingest_message(m)
    assert ev.host_violations() == []
    kinds, a, b = ev.tape()
    ids = wl.output_wire_ids()
    assert len(kinds) == wl.W * (wl.D + 1) and all(kinds[i] in (1, 2) for i in ids)
    batch = inst.shape[0]
    i0 = np.ascontiguousarray(inst[:, :wl.n_instance0])
    _, _, out = oracle_lib.opt_eval_outputs(kinds, a, b, ev.constants(), wl.mod_le, i0.tobytes(), wl.n_instance0,
                                            np.ascontiguousarray(wit).tobytes(), wl.n_witness, wl.width, batch,
                                            threads or min(os.cpu_count() or 1, 64), ids)
    ev.close()
    return out


def check_against_all_lanes_fixture(fx, outputs, lane_offset=0):
    """outputs[lane] (as lane_hash takes them) against the committed per-lane hashes; lanes beyond the fixture fail"""
    hashes = fx['hashes']
    for lane in range(len(outputs)):
        g = lane + lane_offset
        assert g < len(hashes), 'lane %d is beyond the %d lanes of the fixture' % (g, len(hashes))
        assert lane_hash(outputs[lane]) == hashes[g], 'lane %d: outputs differ from the committed oracle-chain hash' % g
