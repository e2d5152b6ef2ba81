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


def flatten(ev, modulus_le, boolean=False):
    """ev: a zkinterface_ir_amd.Evaluator that has ingested the relation(s).
    Returns (relation_bytes, positions) with positions = {'instance': [...], 'witness': [...]}."""
    kinds, a, b = ev.tape()
    consts = ev.constants()
    wire_of = np.full(len(kinds), -1, dtype=np.int64)
    gates, positions, nxt = [], {'instance': [], 'witness': []}, 0
    for i, k in enumerate(kinds):
        k, x, y = int(k), int(a[i]), int(b[i])
        if k == 9:
            gates.append(('assert_zero', int(wire_of[x])))
            continue
        out = nxt
        nxt += 1
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
    rel = write_relation(modulus_le, 'boolean' if boolean else 'arithmetic', 'simple', [], gates)
    return rel, positions


def flattened_inputs(modulus_le, instance_values, witness_values, positions):
    """Instance / Witness messages of the flattened statement for one (instance, witness) pair
    (values = lists of little-endian byte strings of the original statement)."""
    return (write_instance(modulus_le, [instance_values[p] for p in positions['instance']]),
            write_witness(modulus_le, [witness_values[p] for p in positions['witness']]))


def flatten_workspace(paths, out_dir):
    """`zki_sieve flatten <paths> --out <dir>` (cli.rs:442-472): record the statement, write the flattened
    relation and its own instance / witness streams through a FilesSink (000_instance / 001_witness /
    002_relation .sieve).  Returns the Evaluator that recorded it."""
    from . import Evaluator
    from .builder import FilesSink
    ev = Evaluator()
    ev.ingest_paths(list(paths))
    v = ev.host_violations()
    if v:
        raise ValueError('; '.join(v))
    boolean = ev.elem_bytes == 1 if ev.n_value_ops else False
    mod_le = ev.modulus_le()
    rel, positions = flatten(ev, mod_le, boolean=boolean)
    fi, fw = flattened_inputs(mod_le, ev.message_values(False), ev.message_values(True), positions)
    sink = FilesSink.new_clean(out_dir)
    sink.push_instance_message(fi)
    sink.push_witness_message(fw)
    sink.push_relation_message(rel)
    sink.close()
    return ev
