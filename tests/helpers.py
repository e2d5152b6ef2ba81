"""Shared helpers for the test tiers."""
import glob
import os

import numpy as np

import circuits
from oracle_lib import OracleRun
from zkinterface_ir_amd import sieve_writer as sw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_EXAMPLES = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', 'ref_examples', '*.sieve')))


def ref_example_buffers():
    """committed reference fixture, in Source order (instance, witness, relation)"""
    return [open(p, 'rb').read() for p in REF_EXAMPLES]


def golden_buffers(name):
    if name == 'ref_examples':
        return ref_example_buffers()
    return list(circuits.golden_case(name))


def le_values(ints, width):
    return b''.join(int(v).to_bytes(width, 'little') for v in ints)


def lane_messages(mod_le, inst_vals, wit_vals, width):
    """Instance + Witness messages of one lane (values as ints)."""
    return [sw.write_instance(mod_le, [int(v).to_bytes(width, 'little') for v in inst_vals]),
            sw.write_witness(mod_le, [int(v).to_bytes(width, 'little') for v in wit_vals])]


def oracle_lane(mod_le, inst_vals, wit_vals, relation_msgs, width, trace=True):
    return OracleRun(buffers=lane_messages(mod_le, inst_vals, wit_vals, width) + list(relation_msgs), trace=trace,
                     width=max(width, 32))


def batch_arrays(inst_rows, wit_rows, width):
    """rows of ints -> contiguous [batch][n][width] uint8 arrays (bytes)."""
    inst = b''.join(le_values(r, width) for r in inst_rows)
    wit = b''.join(le_values(r, width) for r in wit_rows)
    return inst, wit


def working_entries(ev):
    """entries of the device program that do something: a strand pads with no-ops to keep the entries of a dependency chain
    on one wave (csrc/schedule.cpp, levels that need no barrier between them), and `device_ops` counts those too"""
    ops = ev.schedule_dump()[0]
    return int(((ops[:, 1] & 0xFF) != 0).sum())
