"""Validator and Stats beside the Evaluator (`zki_sieve validate | metrics | valid-eval-metrics`,
rust/src/cli.rs:302-363).  Expected values are the ones the reference's own tests assert
(consumers/validator.rs:863-960, consumers/stats.rs:289-345, structs/value.rs:57-64); the randomised cases
compare against the independent restatement in tests/validator_ref.py.  No GPU involved."""
import copy
import io
import json
import random

import pytest

import zkinterface_ir_amd as zk
from circuits import BLS12_381_Q, BN254_R, arith_example_specs, bool_example_specs, emit_spec, lit32
from helpers import ref_example_buffers
from random_circuits import Gen
from validator_ref import GATE_STAT_FIELDS, StatsRef, ValidatorRef


def run(specs, as_prover=True, metrics=True):
    ev = zk.Evaluator()
    ev.set_option('validate', 'prover' if as_prover else 'verifier')
    if metrics:
        ev.set_option('metrics', '1')
    ev.set_option('max_tape_ops', '2000000')
    for s in specs:
        ev.ingest_message(emit_spec(s))
    return ev


def ref_violations(specs, as_prover=True):
    v = ValidatorRef(as_prover)
    for s in specs:
        v.ingest(s)
    return v.get_violations()


# ---- the reference's own assertions ---------------------------------------------------------------------

def test_validator_example_as_prover():  # validator.rs:863-882
    assert run(arith_example_specs()).validator_violations() == []


def test_validator_example_as_verifier():  # validator.rs:884-901
    inst, _wit, rel = arith_example_specs()
    assert run([inst, rel], as_prover=False).validator_violations() == []


def test_validator_violations():  # validator.rs:903-931
    inst, wit, rel = copy.deepcopy(arith_example_specs())
    inst['values'][0] = inst['mod']              # a value too big for the field
    wit['values'].pop()                          # a witness value missing
    rel['mod'] = bytes([10])                     # different headers
    assert run([inst, wit, rel]).validator_violations() == [
        'The instance value [101, 0, 0, 0] cannot be represented in the field specified in Header (101 >= 101).',
        'The field_characteristic field is not consistent across headers.',
        'Not enough Witness value to consume.',
    ]


def test_validator_free_violations():  # validator.rs:933-960
    inst, wit, rel = copy.deepcopy(arith_example_specs())
    rel['gates'] += [('free', 1, 2), ('free', 4, None)]
    assert run([inst, wit, rel]).validator_violations() == [
        'The wire 1 is used but was not assigned a value, or has been freed already.',
        'The wire 2 is used but was not assigned a value, or has been freed already.',
        'The wire 4 is used but was not assigned a value, or has been freed already.',
    ]


def test_stats_example():  # stats.rs:289-345
    st = run(arith_example_specs()).stats()
    expected = dict.fromkeys(GATE_STAT_FIELDS, 0)
    expected.update(instance_variables=3, witness_variables=4, constants_gates=1, assert_zero_gates=6, copy_gates=0,
                    add_gates=25, mul_gates=21, add_constant_gates=0, mul_constant_gates=1, variables_freed=51,
                    functions_defined=1, functions_called=20, switches=1, branches=2, for_loops=2,
                    instance_messages=1, witness_messages=1, relation_messages=1)
    mul = dict.fromkeys(GATE_STAT_FIELDS, 0)
    mul['mul_gates'] = 1
    assert st == {'field_characteristic': [101, 0, 0, 0], 'field_degree': 1, 'gate_stats': expected,
                  'functions': {'com.example::mul': [mul, 0, 0]}}


def test_committed_workspace_is_compliant_and_counted():
    """rust/examples/*.sieve (the older example the reference ships): valid-eval-metrics host parts."""
    ev = zk.Evaluator()
    ev.set_option('validate', 'prover')
    ev.set_option('metrics', '1')
    for b in ref_example_buffers():
        ev.ingest_message(b)
    assert ev.validator_violations() == []
    assert ev.host_violations() == []
    gs = ev.stats()['gate_stats']
    assert (gs['instance_variables'], gs['witness_variables'], gs['functions_called'], gs['for_loops']) == (3, 3, 4, 1)
    assert not ev.validator_has_live_wires()


def test_live_wire_warning_flag():
    """`WARNING: few variables were not freed.` (validator.rs:138-140) is exposed as a flag."""
    assert run(RULE_CASES['too_many_inputs'], metrics=False).validator_has_live_wires()
    assert not run(arith_example_specs(), metrics=False).validator_has_live_wires()
    with pytest.raises(zk.ZkGpuError, match='not enabled'):
        zk.Evaluator().validator_violations()


def test_boolean_example():  # cli.rs:602-624 runs valid-eval-metrics on it
    ev = run(bool_example_specs())
    assert ev.validator_violations() == []
    assert ev.stats() == stats_ref(bool_example_specs())


def test_primality():  # value.rs:57-64 + the moduli this repo uses
    def prime_violation(p):
        spec = {'type': 'instance', 'mod': p.to_bytes(max(1, (p.bit_length() + 7) // 8), 'little'), 'values': []}
        return 'The field_characteristic should be a prime.' in run([spec]).validator_violations()
    assert prime_violation(187)
    assert not prime_violation(101)
    for p in (2, 3, 997, 1009, 2 ** 61 - 1, BN254_R, BLS12_381_Q, 2 ** 521 - 1):
        assert not prime_violation(p), p
    for c in (0, 1, 4, 561, 1105, 997 * 1009, 341550071728321, 3825123056546413051, (2 ** 61 - 1) * (2 ** 89 - 1),
              BN254_R * BLS12_381_Q, BN254_R + 2, 2 ** 256 - 1):
        assert prime_violation(c), c
    assert 'The field_characteristic should be > 1' in run(
        [{'type': 'instance', 'mod': bytes([1]), 'values': []}]).validator_violations()


def test_bignum_unit(tmp_path):
    """tests/cpp/test_bignum.cpp: the strong Lucas half of the primality test passes every odd prime below 60000 and,
    among the composites, exactly the published strong Lucas pseudoprimes; is_probably_prime agrees with a sieve."""
    import os
    import subprocess
    from helpers import ROOT
    csrc = os.path.join(ROOT, 'zkinterface-ir_amd', 'csrc')
    exe = str(tmp_path / 'test_bignum')
    subprocess.check_call(['g++', '-std=c++17', '-O2', '-I', csrc, os.path.join(ROOT, 'tests', 'cpp', 'test_bignum.cpp'),
                           os.path.join(csrc, 'sieve', 'bignum.cpp'), '-o', exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and 'bad=0' in r.stdout, r.stdout[-2000:]


# ---- one case per rule, checked against the restatement ------------------------------------------------------

def relation(gates, gateset='arithmetic', features='@function,@for,@switch', functions=(), mod=lit32(101), **kw):
    d = {'type': 'relation', 'mod': mod, 'gateset': gateset, 'features': features, 'functions': list(functions),
         'gates': gates}
    d.update(kw)
    return d


def inputs(n_inst, n_wit, mod=lit32(101), **kw):
    a = {'type': 'instance', 'mod': mod, 'values': [lit32(1)] * n_inst}
    b = {'type': 'witness', 'mod': mod, 'values': [lit32(2)] * n_wit}
    a.update(kw)
    b.update(kw)
    return [a, b]


RULE_CASES = {
    'gate_not_allowed': inputs(1, 1) + [relation([('instance', 0), ('witness', 1), ('xor', 2, 0, 1), ('not', 3, 2),
                                                  ('and', 4, 3, 3)])],
    'arith_gate_in_boolean': [{'type': 'instance', 'mod': bytes([2]), 'values': [bytes([1])]},
                              relation([('instance', 0), ('add', 1, 0, 0), ('mulc', 2, 1, bytes([1])),
                                        ('addc', 3, 2, bytes([1])), ('mul', 4, 3, 3)], gateset='boolean', mod=bytes([2]))],
    'boolean_needs_char_2': inputs(1, 0) + [relation([('instance', 0), ('not', 1, 0)], gateset='boolean')],
    'mixed_gateset': inputs(1, 0) + [relation([('instance', 0)], gateset='arithmetic,boolean')
                                     | {'gateset': '@add,@addc,@mul,@mulc,@xor,@and,@not'}],
    'feature_not_allowed': inputs(1, 0) + [relation(
        [('instance', 0), ('call', 'f', [1], [0]), ('anoncall', [2], [1], 0, 0, [('copy', 0, 1)]),
         ('switch', 2, [3], [bytes([1])], [('anon', [2], 0, 0, [('copy', 0, 1)])]),
         ('for', 'i', 0, 1, [(4, 5)], ('anon', [('add', ('name', 'i'), ('const', 4))], [('const', 3)], 0, 0, [('copy', 0, 1)]))],
        features='simple', functions=[('f', 1, 1, 0, 0, [('copy', 0, 1)])])],
    'ssa_and_undefined': inputs(1, 1) + [relation([('instance', 0), ('witness', 0), ('add', 0, 5, 6), ('copy', 7, 8),
                                                   ('assert_zero', 9), ('constant', 0, bytes([3]))])],
    'constant_out_of_field': inputs(1, 0) + [relation([('instance', 0), ('constant', 1, bytes([101])),
                                                       ('addc', 2, 0, bytes([200])), ('mulc', 3, 0, bytes([0, 1])),
                                                       ('constant', 4, b'')])],
    'free_rules': inputs(2, 0) + [relation([('instance', 0), ('instance', 1), ('free', 1, 0), ('free', 0, 1),
                                            ('free', 0, None), ('free', 3, 3)])],
    'bad_ranges': inputs(1, 0) + [relation([('instance', 0), ('call', 'f', [(3, 2)], [0]),
                                            ('anoncall', [(5, 5)], [0], 0, 0, [('copy', 0, 1)])],
                                           functions=[('f', 1, 1, 0, 0, [('copy', 0, 1)])])],
    'call_arity_and_unknown': inputs(1, 0) + [relation([('instance', 0), ('call', 'f', [1, 2], [0]), ('call', 'f', [3], [0, 0]),
                                                        ('call', 'g', [4], [0])],
                                                       functions=[('f', 1, 1, 0, 0, [('copy', 0, 1)])])],
    'function_names': inputs(0, 0) + [relation([], functions=[
        ('ok.name::x_1', 0, 0, 0, 0, []), ('1bad', 0, 0, 0, 0, []), ('a..b', 0, 0, 0, 0, []), ('a:b', 0, 0, 0, 0, []),
        (' padded ', 0, 0, 0, 0, []), ('dup', 0, 0, 0, 0, []), ('dup', 0, 0, 0, 0, []), ('a.b::', 0, 0, 0, 0, []),
        ('café', 0, 0, 0, 0, []), ('a-b', 0, 0, 0, 0, [])])],
    'function_body_rules': inputs(0, 0) + [relation([], functions=[
        ('leaves_inputs', 1, 1, 2, 1, [('instance', 0)]),
        ('no_output', 2, 1, 0, 0, [('copy', 0, 2)]),
        ('uses_outer_iterator', 1, 0, 0, 0, [('constant', 0, bytes([1]))])])],
    'switch_rules': inputs(3, 2) + [relation([
        ('instance', 0),
        ('switch', 0, [1], [bytes([1]), bytes([1]), bytes([200])],
         [('anon', [0], 1, 0, [('instance', 0)]), ('anon', [0], 0, 1, [('witness', 0)])]),
        ('switch', 0, [2], [], []),
        ('switch', 0, [], [], []),
        ('switch', 0, [3], [bytes([4]), bytes([5])], [('call', 'f', [0]), ('anon', [0], 2, 0, [('instance', 0), ('instance', 2)])]),
    ], functions=[('f', 1, 1, 1, 1, [('instance', 2), ('witness', 3), ('add', 0, 2, 3)])])],
    'for_rules': inputs(0, 4) + [relation([
        ('witness', 0),
        ('for', 'i', 5, 2, [], ('anon', [], [], 0, 0, [])),
        ('for', 'i', 1, 2, [(1, 2)],
         ('anon', [('name', 'i')], [('sub', ('name', 'i'), ('const', 1))], 0, 1, [('witness', 2), ('add', 0, 1, 2)])),
        ('for', 'bad name', 3, 3, [3], ('anon', [('name', 'bad name')], [], 0, 1, [('witness', 0)])),
        ('for', 'j', 4, 5, [(4, 6)],
         ('anon', [('name', 'j')], [], 0, 0,
          [('for', 'j', 0, 0, [], ('anon', [], [], 0, 0, [])), ('constant', 0, bytes([1]))])),
    ])],
    'too_many_inputs': inputs(3, 2) + [relation([('instance', 0), ('witness', 1)])],
    'not_enough_inputs': inputs(0, 0) + [relation([('instance', 0), ('witness', 1), ('instance', 2)])],
    'header_rules': [
        {'type': 'instance', 'mod': lit32(100), 'degree': 2, 'version': 'v1', 'values': []},
        {'type': 'witness', 'mod': lit32(100), 'degree': 3, 'version': '1.0.0', 'values': []},
        relation([], mod=lit32(101), degree=2, version='v1')],
    'version_patterns': None,  # expanded below
    'big_field_messages': [
        {'type': 'instance', 'mod': BLS12_381_Q.to_bytes(48, 'little'),
         'values': [(BLS12_381_Q + 12345).to_bytes(48, 'little'), (2 ** 400 + 7).to_bytes(51, 'little'),
                    (BLS12_381_Q - 1).to_bytes(48, 'little')]},
        relation([('instance', 0), ('instance', 1), ('instance', 2)], mod=BLS12_381_Q.to_bytes(48, 'little'))],
}
del RULE_CASES['version_patterns']


@pytest.mark.parametrize('name', sorted(RULE_CASES))
def test_rule(name):
    specs = RULE_CASES[name]
    got = run(specs, metrics=False).validator_violations()
    want = ref_violations(specs)
    assert got == want
    if name not in ('function_names', 'too_many_inputs'):
        assert want, 'the case is meant to violate something'
    # the verifier-side variant drops the Witness message rules
    v_specs = [s for s in specs if s['type'] != 'witness']
    assert run(v_specs, as_prover=False, metrics=False).validator_violations() == ref_violations(v_specs, as_prover=False)


def test_rule_strings_spot_check():
    """A few full strings, so that both sides being wrong the same way would still be noticed."""
    got = run(RULE_CASES['gate_not_allowed'], metrics=False).validator_violations()
    assert got == ['The gate @xor is not allowed in this circuit.', 'The gate @not is not allowed in this circuit.',
                   'The gate @and is not allowed in this circuit.']
    got = run(RULE_CASES['free_rules'], metrics=False).validator_violations()
    assert got == ['For Free gates, last WireId (0) must be strictly greater than first WireId (1).',
                   'The wire 0 is used but was not assigned a value, or has been freed already.',
                   'The variable 0 is being freed, but was not defined previously, or has been already freed'][:1] + got[1:]
    assert 'The wire 3 is used but was not assigned a value, or has been freed already.' in got
    got = run(RULE_CASES['switch_rules'], metrics=False).validator_violations()
    assert 'Gate::Switch: The number of cases value does not match the number of branches.' in got
    assert 'Gate::Switch: The cases values contain duplicates.' in got
    assert ('The Gate::Switch case value: 200 cannot be represented in the field specified in Header (200 >= 101).') in got
    assert 'Switch: no case given while non-empty list of output wires.' in got
    got = run(RULE_CASES['for_rules'], metrics=False).validator_violations()
    assert got[0] == 'In a For loop, the end value (2) must be strictly greater than the start value (5).'
    assert 'Iterator already used in this context.' in got
    assert any(g.startswith('The iterator name (bad name) should match the following format (^[a-zA-Z_]') for g in got)
    got = run(RULE_CASES['big_field_messages'], metrics=False).validator_violations()
    assert got[0].endswith('(%d >= %d).' % (BLS12_381_Q + 12345, BLS12_381_Q))
    assert got[1].endswith('(%d >= %d).' % (2 ** 400 + 7, BLS12_381_Q))
    assert len(got) == 2


@pytest.mark.parametrize('version,ok', [
    ('1.0.0', True), (' 1.0.0\n', True), ('10.20.30', True), ('1a2b3', True), ('12345', True), ('1.0', False),
    ('1..0', False), ('', False), ('v1.0.0', False), ('1.0.0-rc1', False), ('1.0.0.0', False), ('1\n2.3', False),
    ('1234', False), ('1.2.x', False)])
def test_version_pattern(version, ok):
    """`^\\d+.\\d+.\\d+$` with an unescaped dot (validator.rs:23), applied to the trimmed string (:193)."""
    spec = {'type': 'instance', 'mod': lit32(101), 'version': version, 'values': []}
    bad = 'The profile version should match the following format <major>.<minor>.<patch>.' in \
        run([spec], metrics=False).validator_violations()
    assert bad == (not ok)
    assert ref_violations([spec]).count('The profile version should match the following format <major>.<minor>.<patch>.') == (not ok)


# ---- randomised: valid circuits are compliant, damaged ones give the same list as the restatement ---------

def stats_ref(specs):
    s = StatsRef()
    for m in specs:
        s.ingest(m)
    return s.as_dict()


def random_statement(seed, boolean):
    g = Gen(seed, 2 if boolean else random.Random(seed).choice([101, 65521, BN254_R]), boolean=boolean)
    g.relation(n_top=10)
    rows_i, rows_w = g.lane_inputs(1, seed)
    width = max(1, (g.p.bit_length() + 7) // 8)
    mod = g.spec['mod']
    inst = {'type': 'instance', 'mod': mod, 'values': [v.to_bytes(width, 'little') for v in rows_i[0]]}
    wit = {'type': 'witness', 'mod': mod, 'values': [v.to_bytes(width, 'little') for v in rows_w[0]]}
    return [inst, wit, g.spec]


def damage(specs, r):
    """Small structural edits on the tuple form: the result still encodes, but usually breaks some rule."""
    specs = copy.deepcopy(specs)
    rel = specs[2]

    def all_gate_lists(gates, acc):
        acc.append(gates)
        for g in gates:
            if g[0] == 'anoncall':
                all_gate_lists(g[5], acc)
            elif g[0] == 'switch':
                for br in g[4]:
                    if br[0] == 'anon':
                        all_gate_lists(br[4], acc)
            elif g[0] == 'for' and g[5][0] == 'anon':
                all_gate_lists(g[5][5], acc)
        return acc
    lists = all_gate_lists(rel['gates'], [])
    for f in rel['functions']:
        all_gate_lists(f[5], lists)
    lists = [l for l in lists if l]
    for _ in range(r.randrange(1, 4)):
        kind = r.randrange(8)
        gl = r.choice(lists)
        i = r.randrange(len(gl))
        g = gl[i]
        if kind == 0:
            del gl[i]
        elif kind == 1:
            gl.insert(i, g)  # duplicate: SSA
        elif kind == 2 and g[0] in ('add', 'mul', 'and', 'xor', 'copy', 'not'):
            gl[i] = g[:2] + (g[2] + r.randrange(1, 40),) + g[3:]
        elif kind == 3 and specs[0]['values']:
            specs[0]['values'].pop()
        elif kind == 4:
            specs[1]['values'].append(specs[2]['mod'])
        elif kind == 5:
            rel['features'] = r.choice(['simple', '@for', '@function,@switch', '@for,@switch'])
        elif kind == 6:
            rel['gateset'] = r.choice(['@add', '@mul,@mulc', '@xor', '@and,@not', 'boolean', 'arithmetic'])
        elif kind == 7 and len(gl) > 1:
            j = r.randrange(len(gl))
            gl[i], gl[j] = gl[j], gl[i]
    return specs


@pytest.mark.parametrize('boolean', [False, True])
def test_random_circuits_validate_and_count(boolean):
    r = random.Random(77 + boolean)
    n_damaged_with_violations = 0
    for seed in range(60):
        specs = random_statement(1000 * boolean + seed, boolean)
        ev = run(specs)
        assert ev.validator_violations() == [], seed
        assert ev.stats() == stats_ref(specs), seed
        for _ in range(3):
            bad = damage(specs, r)
            try:
                ev = run(bad)
            except zk.ZkGpuError as e:
                # iterator panics etc. abort the reference run too; the restatement raises as well
                with pytest.raises(Exception):
                    ref_violations(bad)
                continue
            want = ref_violations(bad)
            assert ev.validator_violations() == want, (seed, bad)
            assert ev.stats() == stats_ref(bad), seed
            n_damaged_with_violations += bool(want)
    assert n_damaged_with_violations > 60


# ---- the three-part report --------------------------------------------------------------------------------------

def test_stats_json_text_is_serde_pretty():
    text = run(arith_example_specs()).stats_json()
    assert text.startswith('{\n  "field_characteristic": [\n    101,\n    0,\n    0,\n    0\n  ],\n  "field_degree": 1,\n'
                           '  "gate_stats": {\n    "instance_variables": 3,\n')
    assert text.endswith('      0,\n      0\n    ]\n  }\n}')
    assert list(json.loads(text)['gate_stats']) == GATE_STAT_FIELDS
    empty = zk.Evaluator()
    empty.set_option('metrics', '1')
    assert json.loads(empty.stats_json()) == {'field_characteristic': [], 'field_degree': 0,
                                              'gate_stats': dict.fromkeys(GATE_STAT_FIELDS, 0), 'functions': {}}
    assert '"field_characteristic": [],' in empty.stats_json() and '"functions": {}' in empty.stats_json()


def test_unparsable_message_fails_the_run():
    """`let msg = msg?;` (cli.rs:345-346): no report when a message does not decode."""
    ev = zk.Evaluator()
    ev.set_option('validate', 'prover')
    good = emit_spec(arith_example_specs()[0])
    broken = good[:4] + bytes(len(good) - 4)
    with pytest.raises(zk.ZkGpuError):
        ev.ingest_message(broken)


def test_validator_keeps_going_after_the_evaluator_latched():
    """All three consumers see every message (cli.rs:344-349); only the Evaluator ignores messages after its
    first error (evaluator.rs:213-222)."""
    inst, wit, rel = copy.deepcopy(arith_example_specs())
    first = relation([('assert_zero', 77)], mod=lit32(101))
    ev = run([inst, wit, first, rel])
    assert ev.host_violations() == ['No value given for wire_77']
    assert ev.stats()['gate_stats']['relation_messages'] == 2
    assert ev.stats()['gate_stats']['mul_gates'] == 21
    assert ev.validator_violations() == ref_violations([inst, wit, first, rel])


def test_validator_step_limit():
    inst, wit = inputs(1, 0)
    ev = zk.Evaluator()
    ev.set_option('validate', 'prover')
    ev.set_option('validator_max_steps', '1000')
    ev.set_option('max_tape_ops', '1000')
    ev.ingest_message(emit_spec(inst))
    with pytest.raises(zk.ZkGpuError, match='step limit'):
        ev.ingest_message(emit_spec(relation([('instance', 0), ('free', 0, 2 ** 40)])))


def test_cli_validate_and_metrics(tmp_path):
    from zkinterface_ir_amd import cli
    for k, s in enumerate(arith_example_specs()):
        (tmp_path / ('%03d_%s.sieve' % (k, s['type']))).write_bytes(emit_spec(s))
    err, out = io.StringIO(), io.StringIO()
    assert cli.main(['validate', str(tmp_path)], err=err, out=out) == 0
    assert err.getvalue() == '\nThe statement is COMPLIANT with the specification!\n'
    err, out = io.StringIO(), io.StringIO()
    assert cli.main(['metrics', str(tmp_path)], err=err, out=out) == 0
    assert json.loads(out.getvalue())['gate_stats']['functions_called'] == 20
    specs = copy.deepcopy(arith_example_specs())
    specs[2]['gates'].append(('free', 4, None))
    (tmp_path / '002_relation.sieve').write_bytes(emit_spec(specs[2]))
    err, out = io.StringIO(), io.StringIO()
    assert cli.main(['validate', str(tmp_path)], err=err, out=out) == 1
    assert err.getvalue() == ('\nThe statement is NOT COMPLIANT with the specification!\nViolations:\n'
                              '- The wire 4 is used but was not assigned a value, or has been freed already.\n\n'
                              'Error: Found 1 violations.\n')
