"""CPU tier fuzz: random structured relations (functions / for / switch / nesting / frees),
product host (recording + schedule, interpreted by program_sim) vs the oracle, lane by lane."""
import pytest

import circuits
import program_sim
from helpers import oracle_lane
from random_circuits import Gen
import zkinterface_ir_amd as zk

FIELDS = [(101, False), (circuits.BN254_R, False), (2, True), (2 ** 61 - 1, False), (2, False), (101, True),
          (circuits.BN254_R, True)]


def expected_product_violations(ev, first_fail):
    """what zkgpu_lane_violations composes (capi.cpp) for a lane with this first failing assert"""
    host = ev.host_violations()
    out = [m for m in host if m == 'Did not receive any gate to verify.']
    err = [m for m in host if m != 'Did not receive any gate to verify.']
    if first_fail is not None:
        out.append('Wire_%d (may be weighted) should be 0, while it is not' % int(ev.assert_wires()[first_fail]))
    elif err:
        out.append(err[0])
    return out


@pytest.mark.parametrize('seed', range(56))
def test_random_relation_against_oracle(seed):
    p, boolean = FIELDS[seed % len(FIELDS)]
    g = Gen(seed, p, boolean)
    rel, mod_le = g.relation()
    rows_i, rows_w = g.lane_inputs(4, seed + 1000)
    ev = zk.Evaluator()
    ev.declare_inputs(g.n_inst, g.n_wit)
    ev.ingest_message(rel)
    if not ev.n_value_ops and ev.host_violations():
        # nothing recorded (e.g. error in the very first gate): only compare the text
        ref = oracle_lane(mod_le, rows_i[0], rows_w[0], [rel], 32)
        assert ev.host_violations() == ref.violations
        return
    for retain in (True, False):
        ev2 = zk.Evaluator()
        ev2.declare_inputs(g.n_inst, g.n_wit)
        ev2.ingest_message(rel)
        ev2.finalize(retain_all=retain)
        ops, launches, consts, slot_of = ev2.schedule_dump()
        info = ev2.schedule_info()
        kinds, _, _ = ev2.tape()
        for lane in range(4):
            ref = oracle_lane(mod_le, rows_i[lane], rows_w[lane], [rel], 32)
            slots, ff, noncanon = program_sim.simulate(ops, launches, consts, info['words_per_const'],
                                                       info['slots'], p, rows_i[lane], rows_w[lane],
                                                       shuffle_seed=seed, modes=(ev2.input_modes(False), ev2.input_modes(True)))
            assert not noncanon
            assert expected_product_violations(ev2, ff) == ref.violations, (seed, lane)
            if retain:
                vals = [program_sim.from_device_form(slots[slot_of[i]], p, info['words_per_const'])
                        for i in range(len(kinds)) if kinds[i] != 9]
                rv = ref.trace_values()
                assert vals[:len(rv)] == rv, (seed, lane)
