"""CPU tier: the program of the LDS-resident GF(2) kernel (csrc/lds_program.cpp, layout csrc/device/lds_layout.hpp) is
host work -- it is built here for every block size the kernel is instantiated for and INTERPRETED in numpy, 32
witnesses per word like the kernel: the verdicts must be the oracle's / the CPU checker's, and the structural
promises the kernel's hand-counted waits rest on must hold (whole rows, a level never reads what it writes, the
rows of a level are and-rows, one split row, xor-rows)."""
import numpy as np
import pytest

import cpu_checkers
import zkinterface_ir_amd as zk
from helpers import golden_buffers, oracle_lane
from oracle_lib import OracleRun
from zkinterface_ir_amd import workloads

ROW = 2048
K_COPY, K_CONST, K_INSTANCE, K_WITNESS, K_ASSERT, K_AND, K_XOR, K_NOT = 5, 6, 7, 8, 9, 10, 11, 12
NO_FAIL = 0xFFFFFFFF


def pack32(bits):
    """bits[lane][pos] (0/1, <= 32 lanes) -> u32 word per position"""
    lanes = bits.shape[0]
    return (bits.astype(np.uint32) << np.arange(lanes, dtype=np.uint32)[:, None]).sum(axis=0, dtype=np.uint64).astype(np.uint32)


def interpret(prog, consts, inst_bits, wit_bits):
    """-> first failing assert sequence per lane (32 lanes at most); checks the structural promises on the way"""
    lanes = inst_bits.shape[0] if inst_bits.size else wit_bits.shape[0]
    valid = np.uint32((1 << lanes) - 1) if lanes < 32 else np.uint32(0xFFFFFFFF)
    words = prog['table_words']
    T = np.zeros(words, dtype=np.uint32)
    real = words - 34
    T[real + 33] = 0xFFFFFFFF
    pi = pack32(inst_bits) if inst_bits.size else np.zeros(0, np.uint32)
    pw = pack32(wit_bits) if wit_bits.size else np.zeros(0, np.uint32)
    first = np.full(32, NO_FAIL, dtype=np.uint64)
    rows, blocks, ops8, br = prog['rows'], prog['blocks'], prog['ops8'], prog['block_rows']

    def exec_entry(e):
        dst, a, b, kind = (int(x) for x in e)
        if kind == K_XOR: T[dst] = T[a] ^ T[b]
        elif kind == K_AND: T[dst] = T[a] & T[b]
        elif kind == K_NOT: T[dst] = ~T[a]
        elif kind == K_COPY: T[dst] = T[a]
        elif kind == K_CONST: T[dst] = 0xFFFFFFFF if consts[a] else 0
        elif kind == K_INSTANCE: T[dst] = pi[a | (b << 16)]
        elif kind == K_WITNESS: T[dst] = pw[a | (b << 16)]
        elif kind == K_ASSERT:
            nz = int(T[a] & valid)
            seq = dst | (b << 16)
            for lane in range(32):
                if (nz >> lane) & 1:
                    first[lane] = min(first[lane], seq)

    # a level = everything up to a barrier (a block with bit 4, a generic chunk with bit 8): it may hold blocks AND
    # generic chunks (a level's inputs / constants sit between its copy rows and its and / xor rows)
    level_reads, level_writes = set(), set()

    def end_of_level():
        assert not (level_reads & level_writes), 'a level reads a slot it writes: its ops are not independent'
        level_reads.clear()
        level_writes.clear()

    for first_w, nrows, flags, run in prog['chunks']:
        first_w, nrows, flags, run = int(first_w), int(nrows), int(flags), int(run)
        if flags & (1 << 10):
            for blk in range(first_w, first_w + run):
                desc, off = int(blocks[blk, 0]), int(blocks[blk, 1])
                n = desc & 15
                assert 1 <= n <= br and off % 12 == 0
                # `n_and` and-rows, then the split row (its first `split` ops are `and`), then xor-rows (lds_layout.hpp)
                n_and, split = (desc >> 5) & 15, (desc >> 9) & 2047
                shape = (flags >> 11) & 15     # of the run: full blocks with that many and-rows, or 15 = anything
                assert shape == 15 or (n == br and n_and == shape), 'the code of the run would execute other gates'
                assert shape != 15 or n < br or n_and > 12
                assert n_and <= n and split < ROW and (n_and < n or split == 0), 'a split behind the last row of the block'
                assert (off // 2) + br * ROW * 3 <= len(rows), 'a block fetches block_rows rows: they must lie in the stream'
                for r in range(n):
                    rec = rows[off // 2 + r * ROW * 3: off // 2 + (r + 1) * ROW * 3].reshape(ROW, 3).astype(np.int64)
                    dst, a, b = rec[:, 0].copy(), rec[:, 1], rec[:, 2]
                    # the even op of a thread names the slot PAIR (dst / 2): the thread stores its two results with one
                    # ds_write_b64, so the odd op's slot must be the other half of that pair
                    dst[0::2] *= 2
                    assert np.array_equal(dst[1::2], dst[0::2] + 1), 'the two results of a thread are not one aligned slot pair'
                    assert dst.max() < real + 32 and a.max() < words and b.max() < words
                    assert real % 2 == 0, 'the scratch slots start at an even slot'
                    real_dst = dst[dst < real]
                    assert len(np.unique(real_dst)) == len(real_dst), 'two ops of a row write one slot'
                    level_reads.update(a[a < real].tolist())
                    level_reads.update(b[b < real].tolist())
                    assert not (level_writes & set(real_dst.tolist())), 'two rows of a level write one slot'
                    level_writes.update(real_dst.tolist())
                    x, y = T[a], T[b]                      # all reads of the row before its writes, like the kernel
                    is_and = np.arange(ROW) < (ROW if r < n_and else split if r == n_and else 0)
                    T[dst] = np.where(is_and, x & y, x ^ y)
                    T[real + 32] = 0
                    T[real + 33] = 0xFFFFFFFF              # (padding ops only ever write the 32 scratch slots)
                if desc & 16:
                    end_of_level()
        elif flags & (1 << 9) and flags & (1 << 15):
            # a run of narrow levels walked by ONE wave: `nrows` packets of 64 entries, lane l executing entry l of a packet
            # (all its reads before its writes: the 64 lanes run in lockstep), packet after packet with no barrier
            assert flags & (1 << 8)
            end_of_level()
            for pk in range(nrows):
                packet = ops8[first_w + 64 * pk: first_w + 64 * (pk + 1)]
                reads, writes = set(), []
                for dst, a, b, kind in ((int(x) for x in e) for e in packet):
                    if kind in (K_XOR, K_AND, K_NOT, K_COPY, K_ASSERT):
                        reads.add(a)
                        if kind in (K_XOR, K_AND):
                            reads.add(b)
                    if kind in (K_XOR, K_AND, K_NOT, K_COPY, K_CONST, K_INSTANCE, K_WITNESS):
                        assert dst < real + 32, 'an entry writes a constant slot'
                        if dst < real:           # (the padding entries write the scratch slots, any number of them one slot)
                            writes.append(dst)
                assert len(set(writes)) == len(writes), 'two entries of a packet write one slot'
                assert not (reads & set(writes)), 'a packet reads a slot it writes: its entries are not independent'
                for e in packet:
                    exec_entry(e)
        elif flags & (1 << 9):
            assert flags & (1 << 8)
            end_of_level()                                 # (the chunk before ended its level)
            for e in ops8[first_w: first_w + nrows]:       # one thread, in order: a dependent segment
                exec_entry(e)
        else:
            for e in ops8[first_w: first_w + nrows * ROW]:
                dst, a, b, kind = (int(x) for x in e)
                if kind in (K_XOR, K_AND, K_NOT, K_COPY, K_ASSERT):
                    level_reads.add(a)
                    if kind in (K_XOR, K_AND):
                        level_reads.add(b)
                if kind in (K_XOR, K_AND, K_NOT, K_COPY, K_CONST, K_INSTANCE, K_WITNESS) and dst < real:
                    assert dst not in level_writes, 'two ops of a level write one slot'
                    level_writes.add(dst)
                exec_entry(e)
            if flags & (1 << 8):
                end_of_level()
    assert not level_reads and not level_writes, 'the program ends inside a level'
    return first[:lanes]


def session(msgs, n_instance, n_witness):
    ev = zk.Evaluator()
    ev.declare_inputs(n_instance, n_witness)
    for m in msgs:
        ev.ingest_message(m)
    ev.finalize()
    return ev


@pytest.mark.parametrize('block_rows', [0, 4, 6, 8, 9, 10, 12])
def test_lds_program_interpreted_matches_the_cpu_checker(block_rows):
    wl = workloads.BoolLayered(W=4608, D=5, n_instance0=64, n_out=40, seed=0x1D5 + block_rows)   # 5 rows per level
    batch = 23
    inst, wit = wl.inputs(batch)
    outs = cpu_checkers.bool_layered_outputs(wl, inst, wit)
    inst = inst.copy()
    wl.set_expected_outputs(inst, outs, corrupt_every=0)
    want = np.full(batch, NO_FAIL, dtype=np.uint64)
    for lane in range(0, batch, 2):                      # damage one expected output on every other lane
        inst[lane, wl.n_instance0 + lane % wl.n_out, 0] ^= 1
        want[lane] = lane % wl.n_out
    ev = session(wl.relation_messages(), wl.n_instance, wl.n_witness)
    prog = ev.lds_program(block_rows)
    assert prog['block_rows'] == (block_rows or prog['block_rows']) and prog['block_rows'] in (4, 6, 8, 9, 10, 12)
    if block_rows == 0:
        assert prog['block_rows'] == 4                   # levels of 4 and 5 rows: 4-row blocks fetch least (5 + 5 + 10 + ...)
    _, _, consts, _ = ev.schedule_dump()
    got = interpret(prog, consts, inst[:, :, 0], wit[:, :, 0])
    assert got.tolist() == want.tolist()


@pytest.mark.parametrize('mix', [(100, 0), (0, 100), (0, 0), (97, 3), (3, 90), (50, 50)])
def test_row_sequence_of_every_gate_mix(mix):
    """the row ops of a level are one sequence, its `and` ops first: all `and` (the split at the end of the last row, the
    padding ops behind it xor-padded), no `and` at all (split 0), nothing but `not`, and mixes whose split falls anywhere"""
    wl = workloads.BoolLayered(W=5000, D=4, n_instance0=32, n_out=24, seed=0x317 + mix[0], mix=mix)
    batch = 17
    inst, wit = wl.inputs(batch)
    outs = cpu_checkers.bool_layered_outputs(wl, inst, wit)
    inst = inst.copy()
    wl.set_expected_outputs(inst, outs, corrupt_every=0)
    want = np.full(batch, NO_FAIL, dtype=np.uint64)
    for lane in range(0, batch, 3):
        inst[lane, wl.n_instance0 + lane % wl.n_out, 0] ^= 1
        want[lane] = lane % wl.n_out
    ev = session(wl.relation_messages(), wl.n_instance, wl.n_witness)
    _, _, consts, _ = ev.schedule_dump()
    splits = set()
    for block_rows in (4, 8, 12):
        prog = ev.lds_program(block_rows)
        assert interpret(prog, consts, inst[:, :, 0], wit[:, :, 0]).tolist() == want.tolist(), (mix, block_rows)
        splits |= {(int(d) >> 9) & 2047 for d in prog['blocks'][:, 0]}
    if mix[0] in (0,):
        assert splits == {0}
    if mix == (50, 50):
        assert len(splits) > 2      # the boundary moves from level to level


@pytest.mark.parametrize('name', ['bool_correct', 'bool_incorrect'])
def test_lds_program_of_the_reference_examples(name):
    """the reference's Boolean example (functions, a for loop, a switch): mostly sequential segments and short rows"""
    bufs = golden_buffers(name)
    ev = zk.Evaluator.from_messages(bufs)
    ev.finalize()
    ref = OracleRun(buffers=bufs)
    inst = np.array([[v[0] if v else 0 for v in ev.message_values(False)]], dtype=np.uint8)
    wit = np.array([[v[0] if v else 0 for v in ev.message_values(True)]], dtype=np.uint8)
    _, _, consts, _ = ev.schedule_dump()
    got = interpret(ev.lds_program(), consts, inst, wit)
    assert (int(got[0]) == NO_FAIL) == (ref.violations == [])


@pytest.mark.parametrize('seed', range(16))
def test_lds_program_of_random_structured_boolean_relations(seed):
    """functions / for / switch / frees over GF(2) (the generator of the CPU- and GPU-tier fuzz): mostly sequential
    segments and short rows, copies and constants -- which lanes fail, and at which assert first, against the oracle"""
    from random_circuits import Gen
    g = Gen(1000 + seed, 2, True)
    rel, mod_le = g.relation(n_top=14)
    lanes = 9
    rows_i, rows_w = g.lane_inputs(lanes, seed + 77)
    ev = zk.Evaluator()
    ev.declare_inputs(g.n_inst, g.n_wit)
    ev.ingest_message(rel)
    ev.finalize()
    inst = np.array(rows_i, dtype=np.uint8).reshape(lanes, -1)
    wit = np.array(rows_w, dtype=np.uint8).reshape(lanes, -1)
    _, _, consts, _ = ev.schedule_dump()
    got = interpret(ev.lds_program([0, 4, 12][seed % 3]), consts, inst, wit)
    asserts = ev.assert_wires()
    for lane in range(lanes):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32, trace=False)
        if not ref.violations:
            assert int(got[lane]) == NO_FAIL, (seed, lane)
        else:
            assert int(got[lane]) != NO_FAIL, (seed, lane)
            # the reference names the first failing wire
            assert ref.violations[0].startswith('Wire_%d ' % asserts[int(got[lane])]), (seed, lane)


def test_lds_program_refuses_a_field_that_is_not_gf2():
    ev = zk.Evaluator.from_messages(golden_buffers('arith_101_correct'))
    ev.finalize()
    with pytest.raises(zk.ZkGpuError):
        ev.lds_program()


@pytest.mark.parametrize('seed', range(12))
@pytest.mark.parametrize('variant', ['retain_all', 'bank_unaware', 'arithmetic_gates'])
def test_lds_program_of_every_schedule_variant(seed, variant):
    """the row format rests on the scheduler's slot PAIRS (one ds_write_b64 per thread and row): they must also hold for the
    schedules the GPU tier's dumps and switches use -- retain_all (every value keeps its slot), bank_aware = 0 -- and for
    relations whose gates are ADD / MUL / ADDC / MULC over GF(2) (lowered to xor / and / not / copy before the runs are
    formed, so that the runs the slots were paired in are the runs of the program)"""
    from random_circuits import Gen
    g = Gen(2000 + seed, 2, variant != 'arithmetic_gates')
    rel, mod_le = g.relation(n_top=14)
    lanes = 5
    rows_i, rows_w = g.lane_inputs(lanes, seed + 99)
    ev = zk.Evaluator()
    if variant == 'bank_unaware':
        ev.set_option('bank_aware', '0')
    ev.declare_inputs(g.n_inst, g.n_wit)
    ev.ingest_message(rel)
    if not ev.n_value_ops:
        return
    ev.finalize(retain_all=variant == 'retain_all')
    inst = np.array(rows_i, dtype=np.uint8).reshape(lanes, -1)
    wit = np.array(rows_w, dtype=np.uint8).reshape(lanes, -1)
    _, _, consts, _ = ev.schedule_dump()
    got = interpret(ev.lds_program([0, 6, 9][seed % 3]), consts, inst, wit)
    asserts = ev.assert_wires()
    for lane in range(lanes):
        ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32, trace=False)
        if ev.host_violations() and not len(asserts):
            continue
        failing = [v for v in ref.violations if v.startswith('Wire_')]
        if not failing:
            assert int(got[lane]) == NO_FAIL, (seed, lane)
        else:
            assert int(got[lane]) != NO_FAIL and failing[0].startswith('Wire_%d ' % asserts[int(got[lane])]), (seed, lane)


def test_wide_levels_of_every_schedule_variant_keep_their_pairs():
    wl = workloads.BoolLayered(W=2500, D=4, n_instance0=64, n_out=16, seed=0x77)      # runs longer than 64 ops, ragged rows
    batch = 9
    inst, wit = wl.inputs(batch)
    outs = cpu_checkers.bool_layered_outputs(wl, inst, wit)
    inst = inst.copy()
    wl.set_expected_outputs(inst, outs, corrupt_every=4)
    want = [0 if lane % 4 == 0 else NO_FAIL for lane in range(batch)]
    for retain, bank in ((True, '1'), (False, '0'), (False, '1')):
        ev = zk.Evaluator()
        ev.set_option('bank_aware', bank)
        ev.declare_inputs(wl.n_instance, wl.n_witness)
        for m in wl.relation_messages():
            ev.ingest_message(m)
        ev.finalize(retain_all=retain)
        _, _, consts, _ = ev.schedule_dump()
        assert interpret(ev.lds_program(), consts, inst[:, :, 0], wit[:, :, 0]).tolist() == want, (retain, bank)


def test_narrow_levels_run_as_packets_walked_by_one_wave():
    """Real Boolean circuits are thousands of levels of a few dozen gates (the SHA-256 compression: 1.2 * 10^5 gates in 3,900
    levels): levels of fewer than 257 ops are not padded to 2048-op rows with a barrier each but form runs of 64-entry
    packets that one wave walks (lds_layout.hpp kLdsChunkWave).  Four rounds of the compression function: the program's
    verdicts are the oracle's, and the interpreter checks that the entries of a packet are independent."""
    wl = workloads.Sha256Compress(rounds=4)
    msgs = wl.relation_messages()
    lanes = 5
    rng = np.random.default_rng(3)
    wit = rng.integers(0, 2, size=(lanes, 512, 1), dtype=np.uint8)
    inst = np.zeros((lanes, 256, 1), dtype=np.uint8)
    ev = zk.Evaluator()
    ev.declare_inputs(wl.n_instance, wl.n_witness)
    for m in msgs:
        ev.ingest_message(m)
    ev.finalize()
    P = ev.lds_program(0)
    wave_chunks = [c for c in P['chunks'] if int(c[2]) & (1 << 15)]
    assert wave_chunks and sum(int(c[1]) for c in wave_chunks) * 64 > wl.n_gates   # most gates sit in packets
    first = interpret(P, ev.schedule_dump()[2], inst[:, :, 0], wit[:, :, 0])
    # the claimed digest is all zeros: the first assert that fails is the first digest bit that is 1
    for lane in range(lanes):
        ref = oracle_lane(wl.mod_le, [0] * 256, [int(x) for x in wit[lane, :, 0]], msgs, 1, trace=False)
        want = [] if first[lane] == NO_FAIL else ['Wire_%d (may be weighted) should be 0, while it is not' % ev.assert_wires()[int(first[lane])]]
        assert want == ref.violations and ref.violations, lane
    # the row path instead (every level its own padded row): same verdicts
    ev2 = zk.Evaluator()
    ev2.set_option('bool_narrow_width', '3')
    ev2.declare_inputs(wl.n_instance, wl.n_witness)
    for m in msgs:
        ev2.ingest_message(m)
    ev2.finalize()
    assert np.array_equal(interpret(ev2.lds_program(0), ev2.schedule_dump()[2], inst[:, :, 0], wit[:, :, 0]), first)
