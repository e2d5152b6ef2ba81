"""Producer side: sinks and the gate builder, so that circuits can be built against the reference's producer
API and written as `.sieve` workspaces for `evaluate` (GPU) or for the original `zki_sieve` elsewhere.

Mirrors the behaviour (wire numbering, message chunking, error texts) of
  rust/src/producers/sink.rs:7-145            Sink / MemorySink / FilesSink / clean_workspace
  rust/src/producers/builder.rs:36-134        MessageBuilder (flush at max_len = 100 000)
  rust/src/producers/builder.rs:151-385       GateBuilder (+ create_complex_gate, push_function)
  rust/src/producers/builder.rs:409-509       FunctionBuilder
  rust/src/producers/builder.rs:584-720       SwitchBuilder / SwitchParams
  rust/src/producers/build_gates.rs:10-86     BuildGate / BuildComplexGate (no-output forms)
  rust/src/structs/gates.rs:742-854           replace_output_wires
Gates are the tuples of sieve_writer.py; a build gate is the same tuple without its output:
  ('constant', value) ('assert_zero', w) ('copy', w) ('add'|'mul'|'and'|'xor', l, r) ('addc'|'mulc', w, value)
  ('not', w) ('instance', value|None) ('witness', value|None) ('free', first, last|None)
and a complex build gate is ('call', name, input_wirelist) or the object SwitchBuilder.finish() returns.
"""
import os

from . import sieve_writer as sw

ADD, ADDC, MUL, MULC, ARITH = 0x0001, 0x0002, 0x0004, 0x0008, 0x000F
XOR, AND, NOT, BOOL = 0x0100, 0x0200, 0x0400, 0x0700
FUNCTION, FOR, SWITCH, SIMPLE, FOR_FUNCTION_SWITCH = 0x1000, 0x2000, 0x4000, 0x0000, 0x7000
NO_OUTPUT = 2 ** 64 - 1  # build_gates.rs:27
IR_VERSION = '1.0.0'     # structs/mod.rs:36
FILE_EXTENSION = 'sieve'


class BuilderError(Exception):
    """`Err(...)` of the reference's builder, with its message text."""


def create_gateset_string(gateset):  # structs/relation.rs:181-225
    if gateset & ARITH == ARITH:
        return 'arithmetic'
    if gateset & BOOL == BOOL:
        return 'boolean'
    out = ''
    for bit, name in ((ADD, '@add,'), (ADDC, '@addc,'), (MUL, '@mul,'), (MULC, '@mulc,'), (XOR, '@xor,'),
                      (NOT, '@not,'), (AND, '@and,')):
        if gateset & bit:
            out += name
    return out


def create_feature_string(features):  # structs/relation.rs:255-281
    if features & FOR_FUNCTION_SWITCH == 0:
        return 'simple'
    out = ''
    for bit, name in ((FOR, '@for,'), (SWITCH, '@switch,'), (FUNCTION, '@function,')):
        if features & bit:
            out += name
    return out


class Header:  # structs/header.rs:11-35
    def __init__(self, field_characteristic=b'', version=IR_VERSION, field_degree=1):
        self.version, self.field_characteristic, self.field_degree = version, bytes(field_characteristic), field_degree


# ---- sinks (sink.rs) ------------------------------------------------------------------------------------

class MemorySink:
    """Three byte buffers; `buffers()` is `Into<Source>` (instance, witness, relation order, sink.rs:49-57)."""

    def __init__(self):
        self.instance_buffer, self.witness_buffer, self.relation_buffer = bytearray(), bytearray(), bytearray()

    def push_instance_message(self, data):
        self.instance_buffer += data

    def push_witness_message(self, data):
        self.witness_buffer += data

    def push_relation_message(self, data):
        self.relation_buffer += data

    def buffers(self):
        return [bytes(self.instance_buffer), bytes(self.witness_buffer), bytes(self.relation_buffer)]


def has_sieve_extension(path):
    return os.path.splitext(path)[1] == '.' + FILE_EXTENSION


def clean_workspace(workspace):  # sink.rs:133-145
    for f in os.listdir(workspace):
        if has_sieve_extension(f):
            os.remove(os.path.join(workspace, f))


class FilesSink:
    """000_instance.sieve / 001_witness.sieve / 002_relation.sieve in a workspace directory (sink.rs:60-131);
    messages of one type are appended to the same file."""

    def __init__(self, workspace, clean=True):
        os.makedirs(workspace, exist_ok=True)
        if clean:
            clean_workspace(workspace)
        self.workspace = workspace
        self._files = [open(p, 'wb') for p in (self.instance_path(workspace), self.witness_path(workspace),
                                               self.relation_path(workspace))]

    new_clean = classmethod(lambda cls, workspace: cls(workspace, clean=True))
    new_no_cleanup = classmethod(lambda cls, workspace: cls(workspace, clean=False))

    @staticmethod
    def instance_path(workspace):
        return os.path.join(workspace, '000_instance.' + FILE_EXTENSION)

    @staticmethod
    def witness_path(workspace):
        return os.path.join(workspace, '001_witness.' + FILE_EXTENSION)

    @staticmethod
    def relation_path(workspace):
        return os.path.join(workspace, '002_relation.' + FILE_EXTENSION)

    def _push(self, k, data):
        self._files[k].write(data)
        self._files[k].flush()

    def push_instance_message(self, data):
        self._push(0, data)

    def push_witness_message(self, data):
        self._push(1, data)

    def push_relation_message(self, data):
        self._push(2, data)

    def close(self):
        for f in self._files:
            f.close()

    def paths(self):
        return [self.workspace]


# ---- wire-id helpers ---------------------------------------------------------------------------------------

def expand_wirelist(wl):  # structs/wire.rs:178-203
    out = []
    for e in wl:
        if isinstance(e, tuple):
            if e[1] <= e[0]:
                raise BuilderError('In WireRange, last WireId (%d) must be strictly greater than first WireId (%d).' % (e[1], e[0]))
            out.extend(range(e[0], e[1] + 1))
        else:
            out.append(e)
    return out


def wirelist_len(wl):  # structs/wire.rs:221-229
    return sum((e[1] - e[0] + 1) if isinstance(e, tuple) else 1 for e in wl)


def _replace_in_wirelist(wl, old, new):  # structs/wire.rs:233-250
    wires = expand_wirelist(wl)
    if old in wires:
        return [new if w == old else w for w in wires]
    return wl


def replace_output_wires(gates, output_wires):
    """structs/gates.rs:742-854: rename output_wires[i] to i in place; with a For gate present, append
    Copy(i, output_wires[i]) instead; freeing an output wire is an error."""
    if any(g[0] == 'for' for g in gates):
        for i, w in enumerate(output_wires):
            gates.append(('copy', i, w))
        return
    for new, old in enumerate(output_wires):
        def r(w):
            return new if w == old else w
        for k, g in enumerate(gates):
            kind = g[0]
            if kind in ('constant',):
                gates[k] = (kind, r(g[1]), g[2])
            elif kind in ('copy', 'not'):
                gates[k] = (kind, r(g[1]), r(g[2]))
            elif kind in ('add', 'mul', 'and', 'xor'):
                gates[k] = (kind, r(g[1]), r(g[2]), r(g[3]))
            elif kind in ('addc', 'mulc'):
                gates[k] = (kind, r(g[1]), r(g[2]), g[3])
            elif kind in ('instance', 'witness', 'assert_zero'):
                gates[k] = (kind, r(g[1]))
            elif kind == 'free':
                first, last = g[1], g[2]
                if (last is not None and first <= old <= last) or (last is None and first == old):
                    raise BuilderError('It is forbidden to free an output wire !')
            elif kind == 'anoncall':
                gates[k] = (kind, _replace_in_wirelist(g[1], old, new), _replace_in_wirelist(g[2], old, new)) + tuple(g[3:])
            elif kind == 'call':
                gates[k] = (kind, g[1], _replace_in_wirelist(g[2], old, new), _replace_in_wirelist(g[3], old, new))
            elif kind == 'switch':
                branches = []
                for br in g[4]:
                    if br[0] == 'call':
                        branches.append(('call', br[1], _replace_in_wirelist(br[2], old, new)))
                    else:
                        branches.append(('anon', _replace_in_wirelist(br[1], old, new)) + tuple(br[2:]))
                gates[k] = (kind, r(g[1]), _replace_in_wirelist(g[2], old, new), g[3], branches)


def _has_output(bg):  # build_gates.rs:56-62
    return bg[0] not in ('assert_zero', 'free')


def _with_output(bg, out):  # build_gates.rs:33-54
    k = bg[0]
    if k in ('assert_zero', 'free'):
        assert out == NO_OUTPUT
        return tuple(bg)
    if k in ('instance', 'witness'):
        return (k, out)
    return (k, out) + tuple(bg[1:])


def _multiple_alloc(free_id, n):  # builder.rs:243-254
    if n == 0:
        return [], free_id
    if n == 1:
        return [free_id], free_id + 1
    return [(free_id, free_id + n - 1)], free_id + n


class FunctionParams:  # builder.rs:160-218
    def __init__(self, input_count, output_count, instance_count, witness_count):
        self.input_count, self.output_count = input_count, output_count
        self.instance_count, self.witness_count = instance_count, witness_count

    def check(self, name, input_count=None, output_count=None, instance_count=None, witness_count=None):
        for got, have, what in ((input_count, self.input_count, 'inputs'), (output_count, self.output_count, 'outputs'),
                                (instance_count, self.instance_count, 'instances'),
                                (witness_count, self.witness_count, 'witnesses')):
            if got is not None and got != have:
                raise BuilderError('Function %s has %d %s and is called with %d %s.' % (name, have, what, got, what))


def _known(known_functions, name):  # builder.rs:222-230
    if name not in known_functions:
        raise BuilderError('Function %s does not exist !' % name)
    return known_functions[name]


class SwitchGate:
    """BuildComplexGate::Switch(condition, cases, branches, params) (build_gates.rs:70-71)."""

    def __init__(self, condition, cases, branches, output_count, instance_count, witness_count):
        self.condition, self.cases, self.branches = condition, cases, branches
        self.output_count, self.instance_count, self.witness_count = output_count, instance_count, witness_count

    def check(self, instance_count, witness_count):  # SwitchParams::check, builder.rs:677-717
        if instance_count != self.instance_count:
            raise BuilderError('Switch has %d instances and is called with %d instances.' % (self.instance_count, instance_count))
        if witness_count != self.witness_count:
            # the reference prints instance_count in this message (builder.rs:710-713)
            raise BuilderError('Switch has %d witnesses and is called with %d witnesses.' % (self.instance_count, witness_count))


class MessageBuilder:  # builder.rs:36-134
    def __init__(self, sink, header, gateset, features):
        self.sink, self.header = sink, header
        self.gateset, self.features = gateset, features
        self.instance, self.witness = [], []
        self.functions, self.gates = [], []
        self.functions_size = 0
        self.max_len = 100 * 1000

    def push_instance_value(self, v):
        self.instance.append(bytes(v))
        if len(self.instance) == self.max_len:
            self.flush_instance()

    def push_witness_value(self, v):
        self.witness.append(bytes(v))
        if len(self.witness) == self.max_len:
            self.flush_witness()

    def push_gate(self, g):
        self.gates.append(g)
        if len(self.gates) + self.functions_size >= self.max_len:
            self.flush_relation()

    def push_function(self, f):
        self.functions_size += len(f[5])
        self.functions.append(f)
        if len(self.gates) + self.functions_size >= self.max_len:
            self.flush_relation()

    def _kw(self):
        return dict(degree=self.header.field_degree, version=self.header.version)

    def flush_instance(self):
        self.sink.push_instance_message(sw.write_instance(self.header.field_characteristic, self.instance, **self._kw()))
        self.instance = []

    def flush_witness(self):
        self.sink.push_witness_message(sw.write_witness(self.header.field_characteristic, self.witness, **self._kw()))
        self.witness = []

    def flush_relation(self):
        self.sink.push_relation_message(sw.write_relation(
            self.header.field_characteristic, create_gateset_string(self.gateset), create_feature_string(self.features),
            self.functions, self.gates, **self._kw()))
        self.gates, self.functions, self.functions_size = [], [], 0

    def finish(self):
        if self.instance:
            self.flush_instance()
        if self.witness:
            self.flush_witness()
        if self.gates or self.functions:
            self.flush_relation()
        return self.sink


class GateBuilder:  # builder.rs:151-385
    def __init__(self, sink, header, gateset, features):
        self.msg_build = MessageBuilder(sink, header, gateset, features)
        self.known_functions = {}
        self.free_id = 0

    def create_gate(self, gate):
        out = NO_OUTPUT
        if _has_output(gate):
            out, self.free_id = self.free_id, self.free_id + 1
        if gate[0] == 'instance' and len(gate) > 1 and gate[1] is not None:
            self.push_instance_value(gate[1])
        elif gate[0] == 'witness' and len(gate) > 1 and gate[1] is not None:
            self.push_witness_value(gate[1])
        self.msg_build.push_gate(_with_output(gate, out))
        return out

    def create_complex_gate(self, gate, instances=(), witnesses=()):
        if isinstance(gate, SwitchGate):
            gate.check(len(instances), len(witnesses))
            output_count = gate.output_count
        else:
            _, name, input_wires = gate
            params = _known(self.known_functions, name)
            params.check(name, input_count=len(expand_wirelist(input_wires)), instance_count=len(instances),
                         witness_count=len(witnesses))
            output_count = params.output_count
        for v in instances:
            self.msg_build.push_instance_value(v)
        for v in witnesses:
            self.msg_build.push_witness_value(v)
        outs, self.free_id = _multiple_alloc(self.free_id, output_count)
        self.msg_build.push_gate(_complex_with_output(gate, outs))
        return list(outs)

    def new_function_builder(self, name, output_count, input_count):
        return FunctionBuilder(name, output_count, input_count, self.known_functions)

    def new_switch_builder(self, output_count):
        return SwitchBuilder(output_count, self.known_functions)

    def push_witness_value(self, v):
        self.msg_build.push_witness_value(v)

    def push_instance_value(self, v):
        self.msg_build.push_instance_value(v)

    def push_function(self, function):
        name = function[0]
        if name in self.known_functions:
            raise BuilderError('Function %s already exists !' % name)
        self.known_functions[name] = FunctionParams(input_count=function[2], output_count=function[1],
                                                    instance_count=function[3], witness_count=function[4])
        self.msg_build.push_function(function)

    def finish(self):
        return self.msg_build.finish()


def _complex_with_output(gate, outs):  # build_gates.rs:77-85
    if isinstance(gate, SwitchGate):
        return ('switch', gate.condition, list(outs), list(gate.cases), list(gate.branches))
    return ('call', gate[1], list(outs), list(gate[2]))


class FunctionBuilder:  # builder.rs:409-509
    def __init__(self, name, output_count, input_count, known_functions):
        self.name, self.output_count, self.input_count = name, output_count, input_count
        self.gates = []
        self.instance_count = self.witness_count = 0
        self.known_functions = known_functions
        self.free_id = output_count + input_count

    def input_wire_ids(self):
        return list(range(self.output_count, self.output_count + self.input_count))

    def create_gate(self, gate):
        out = NO_OUTPUT
        if _has_output(gate):
            out, self.free_id = self.free_id, self.free_id + 1
        if gate[0] == 'instance':
            self.instance_count += 1
        elif gate[0] == 'witness':
            self.witness_count += 1
        self.gates.append(_with_output(gate, out))
        return out

    def create_complex_gate(self, gate):
        if isinstance(gate, SwitchGate):
            oc, ic, wc = gate.output_count, gate.instance_count, gate.witness_count
        else:
            _, name, input_wires = gate
            params = _known(self.known_functions, name)
            n_in = len(expand_wirelist(input_wires))
            if params.input_count != n_in:
                raise BuilderError('Function %s has %d inputs and is called with %d inputs.' % (name, params.input_count, n_in))
            oc, ic, wc = params.output_count, params.instance_count, params.witness_count
        outs, self.free_id = _multiple_alloc(self.free_id, oc)
        self.witness_count += wc
        self.instance_count += ic
        self.gates.append(_complex_with_output(gate, outs))
        return list(outs)

    def finish(self, output_wires):
        if len(output_wires) != self.output_count:
            raise BuilderError('Function %s should return %d outputs (and not %d)' % (self.name, self.output_count, len(output_wires)))
        replace_output_wires(self.gates, list(output_wires))
        return (self.name, self.output_count, self.input_count, self.instance_count, self.witness_count, list(self.gates))


class SwitchBuilder:  # builder.rs:584-668
    def __init__(self, output_count, known_functions):
        self.output_count = output_count
        self.cases, self.branches = [], []
        self.instance_count = self.witness_count = 0
        self.known_functions = known_functions

    def create_branch_from(self, name, inputs):
        params = _known(self.known_functions, name)
        params.check(name, input_count=wirelist_len(inputs))
        return (('call', name, list(inputs)), params)

    def push_branch(self, branch, case):
        invoke, params = branch
        if self.output_count != params.output_count:
            raise BuilderError('The switch has %d outputs and the branch has %d outputs.' % (self.output_count, params.output_count))
        if bytes(case) in self.cases:
            raise BuilderError('You cannot create a switch with two cases with the same value.')
        self.instance_count = max(self.instance_count, params.instance_count)
        self.witness_count = max(self.witness_count, params.witness_count)
        self.cases.append(bytes(case))
        self.branches.append(invoke)

    def finish(self, condition):
        if len(self.branches) != len(self.cases):
            raise BuilderError('The switch has %d branches and %d cases.' % (len(self.branches), len(self.cases)))
        if not self.branches:
            raise BuilderError('Cannot create an empty switch !')
        return SwitchGate(condition, self.cases, self.branches, self.output_count, self.instance_count, self.witness_count)
