"""Malformed `.sieve` bytes must surface as errors (the reference panics on them, evaluator.rs:193),
never as crashes or hangs: bit flips, truncations and garbage through both FlatBuffers walkers
(product reader and oracle reader).  Each case runs in-process: a segfault would kill pytest."""
import random

import pytest

from helpers import golden_buffers
from oracle_lib import OracleRun
import zkinterface_ir_amd as zk


def _mutations(buf, rng, n):
    out = []
    for _ in range(n):
        b = bytearray(buf)
        kind = rng.randrange(4)
        if kind == 0:      # flip a few bytes (keep the size prefix so the message is actually parsed)
            for _ in range(rng.randrange(1, 6)):
                b[rng.randrange(4, len(b))] = rng.randrange(256)
        elif kind == 1:    # overwrite an aligned 32-bit word with a large offset
            at = rng.randrange(1, len(b) // 4) * 4
            b[at:at + 4] = rng.choice([0xFFFFFFFF, 0x7FFFFFFF, len(b), len(b) * 2, 0]).to_bytes(4, 'little')
        elif kind == 2:    # truncate but patch the size prefix to the new length
            cut = rng.randrange(8, len(b))
            b = b[:cut]
            b[0:4] = (cut - 4).to_bytes(4, 'little')
        else:              # random tail
            at = rng.randrange(8, len(b))
            b[at:] = bytes(rng.randrange(256) for _ in range(len(b) - at))
        out.append(bytes(b))
    return out


@pytest.mark.parametrize('name', ['arith_101_correct', 'bool_correct', 'ref_examples'])
def test_mutated_messages_do_not_crash(name):
    inst, wit, rel = golden_buffers(name)
    rng = random.Random(hash(name) & 0xFFFF)
    n_err = 0
    for which, buf in (('relation', rel), ('instance', inst), ('witness', wit)):
        for bad in _mutations(buf, rng, 150 if which == 'relation' else 40):
            msgs = {'relation': [inst, wit, bad], 'instance': [bad, wit, rel], 'witness': [inst, bad, rel]}[which]
            ev = zk.Evaluator.from_messages(msgs, max_tape_ops=1 << 20)  # a corrupt loop bound stops here
            v = ev.host_violations()
            ref = OracleRun(buffers=msgs, trace=False, max_ops=1 << 20)
            rv = ref.violations
            assert isinstance(v, list) and isinstance(rv, list)
            n_err += bool(v)
            if v:                      # whatever was recorded can still be scheduled
                if ev.n_value_ops:
                    ev.finalize()
    assert n_err > 20  # the mutations do reach the error paths


def test_garbage_and_empty_streams():
    for junk in (b'', b'\x00', b'\x00\x00\x00\x00', b'\xff\xff\xff\xff', bytes(range(256)) * 4,
                 (1 << 31).to_bytes(4, 'little') + b'abcd'):
        ev = zk.Evaluator.from_messages([junk])
        assert ev.host_violations() == OracleRun(buffers=[junk]).violations


def test_self_recursive_function_and_runaway_loop_are_refused():
    from zkinterface_ir_amd import sieve_writer as sw
    mod = bytes([101])
    rec = sw.write_relation(mod, 'arithmetic', '@function', [('f', 1, 0, 0, 0, [('call', 'f', [0], [])])],
                            [('call', 'f', [0], [])])
    v = zk.Evaluator.from_messages([rec]).host_violations()
    assert v == OracleRun(buffers=[rec]).violations and 'nested deeper' in v[0]
    loop = sw.write_relation(mod, 'arithmetic', '@for', [], [
        ('constant', 0, bytes([1])),
        ('for', 'i', 1, 2 ** 40, [(1, 2)], ('anon', [('name', 'i')], [('sub', ('name', 'i'), ('const', 1))], 0, 0,
                                             [('add', 0, 1, 1)]))])
    v = zk.Evaluator.from_messages([loop], max_tape_ops=100000).host_violations()
    assert len(v) == 1 and 'max_tape_ops' in v[0]
    huge = sw.write_relation(mod, 'arithmetic', '@function', [], [('anoncall', [(0, 2 ** 50)], [], 0, 0, [])])
    v = zk.Evaluator.from_messages([huge]).host_violations()
    assert len(v) == 1 and 'more than 2^28 wires' in v[0]
