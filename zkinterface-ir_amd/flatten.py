"""Emit the recorded tape as a flattened SIEVE IR workspace (simple gates only).

Counterpart of the reference's `IRFlattener` (rust/src/consumers/flattening.rs:42-191, the `flatten`
tool of cli.rs:120-152): every value-returning backend call becomes one simple gate whose output wire
is the call's index among the value-returning calls (the numbering of builder.rs:229-233), every
assert_zero an AssertZero gate.  Because Switch branches re-read the same queued values
(evaluator.rs:586-591), the flattened statement has its own instance / witness streams: the i-th
Instance gate reads `positions['instance'][i]` of the original stream."""
import numpy as np

from . import KIND_NAMES
from .sieve_writer import write_instance, write_relation, write_witness

_BINARY = {1: 'add', 2: 'mul', 10: 'and', 11: 'xor'}
_UNARY = {5: 'copy', 12: 'not'}
_CONSTG = {3: 'addc', 4: 'mulc'}


_GATE_BITS = {'add': 0x0001, 'addc': 0x0002, 'mul': 0x0004, 'mulc': 0x0008, 'xor': 0x0100, 'and': 0x0200, 'not': 0x0400}


def flatten(ev, modulus_le, boolean=False, gate_mask=None):
    """ev: a zkinterface_ir_amd.Evaluator that has ingested the relation(s).
    Returns (relation_bytes, positions) with positions = {'instance': [...], 'witness': [...]}.

    gate_mask (structs/relation.rs:15-25 bits) turns the flattener into the reference's ExpandDefinable
    (rust/src/consumers/exp_definable.rs:24-139): a gate the target gateset lacks is rewritten with the ones it has --
    add <-> xor, mul <-> and, addc -> constant + add, mulc -> constant + mul, not -> addc(one) -- and a rewrite that
    needs a missing gate too raises, where the reference panics."""
    kinds, a, b = ev.tape()
    consts = ev.constants()
    wire_of = np.full(len(kinds), -1, dtype=np.int64)
    gates, positions = [], {'instance': [], 'witness': []}
    nxt = [0]

    def has(name):
        return gate_mask is None or (gate_mask & _GATE_BITS[name]) == _GATE_BITS[name]

    def fresh():
        nxt[0] += 1
        return nxt[0] - 1

    def emit_binary(name, x, y):  # exp_definable.rs:60-82,106-128
        swap = {'add': 'xor', 'xor': 'add', 'mul': 'and', 'and': 'mul'}
        if not has(name):
            if not has(swap[name]):
                raise ValueError('Cannot replace %s by %s if %s is not supported.' % (name.upper(), swap[name].upper(), swap[name].upper()))
            name = swap[name]
        out = fresh()
        gates.append((name, out, x, y))
        return out

    def emit_with_constant(name, x, value):  # exp_definable.rs:84-104
        if has(name):
            out = fresh()
            gates.append((name, out, x, value))
            return out
        tmp = fresh()
        gates.append(('constant', tmp, value))
        return emit_binary('add' if name == 'addc' else 'mul', x, tmp)

    for i, k in enumerate(kinds):
        k, x, y = int(k), int(a[i]), int(b[i])
        if k == 9:
            gates.append(('assert_zero', int(wire_of[x])))
            continue
        if gate_mask is not None and k in (1, 2, 10, 11):
            wire_of[i] = emit_binary(_BINARY[k], int(wire_of[x]), int(wire_of[y]))
            continue
        if gate_mask is not None and k in _CONSTG:
            wire_of[i] = emit_with_constant(_CONSTG[k], int(wire_of[x]), consts[y])
            continue
        if gate_mask is not None and k == 12 and not has('not'):  # exp_definable.rs:130-139
            if not has('add'):
                raise ValueError('Cannot replace NOT by ADD if ADD is not supported.')
            wire_of[i] = emit_with_constant('addc', int(wire_of[x]), bytes([1]))
            continue
        out = fresh()
        wire_of[i] = out
        if k in _BINARY:
            gates.append((_BINARY[k], out, int(wire_of[x]), int(wire_of[y])))
        elif k in _UNARY:
            gates.append((_UNARY[k], out, int(wire_of[x])))
        elif k in _CONSTG:
            gates.append((_CONSTG[k], out, int(wire_of[x]), consts[y]))
        elif k == 6:
            gates.append(('constant', out, consts[x]))
        elif k == 7:
            gates.append(('instance', out))
            positions['instance'].append(x)
        elif k == 8:
            gates.append(('witness', out))
            positions['witness'].append(x)
        else:
            raise ValueError(KIND_NAMES.get(k, k))
    if gate_mask is None:
        gateset = 'boolean' if boolean else 'arithmetic'
    else:
        from .builder import create_gateset_string
        gateset = create_gateset_string(gate_mask)
    rel = write_relation(modulus_le, gateset, 'simple', [], gates)
    return rel, positions


def flattened_inputs(modulus_le, instance_values, witness_values, positions):
    """Instance / Witness messages of the flattened statement for one (instance, witness) pair
    (values = lists of little-endian byte strings of the original statement)."""
    return (write_instance(modulus_le, [instance_values[p] for p in positions['instance']]),
            write_witness(modulus_le, [witness_values[p] for p in positions['witness']]))


def parse_gate_set(gateset):
    """structs/relation.rs:144-167"""
    mask = 0
    for sub in gateset.split(','):
        sub = sub.replace(' ', '')
        if sub == 'arithmetic':
            return 0x000F
        if sub == 'boolean':
            return 0x0700
        if sub == '':
            continue
        if sub[:1] != '@' or sub[1:] not in _GATE_BITS:
            raise ValueError('Unable to parse the following gateset: %s' % gateset)
        mask |= _GATE_BITS[sub[1:]]
    return mask


def flatten_workspace(paths, out_dir, gate_mask=None):
    """`zki_sieve flatten <paths> --out <dir>` (cli.rs:442-472) and, with a gate mask, `expand-definable`
    (cli.rs:515-555): record the statement, write the flattened relation and its own instance / witness
    streams through a FilesSink (000_instance / 001_witness / 002_relation .sieve).  Returns the Evaluator
    that recorded it."""
    from . import Evaluator
    from .builder import FilesSink
    ev = Evaluator()
    ev.ingest_paths(list(paths))
    v = ev.host_violations()
    if v:
        raise ValueError('; '.join(v))
    boolean = ev.elem_bytes == 1 if ev.n_value_ops else False
    mod_le = ev.modulus_le()
    rel, positions = flatten(ev, mod_le, boolean=boolean, gate_mask=gate_mask)
    fi, fw = flattened_inputs(mod_le, ev.message_values(False), ev.message_values(True), positions)
    sink = FilesSink.new_clean(out_dir)
    sink.push_instance_message(fi)
    sink.push_witness_message(fw)
    sink.push_relation_message(rel)
    sink.close()
    return ev
