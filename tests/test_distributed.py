"""CPU tier, sharding arithmetic only (gloo, world size 2): lanes are sharded contiguously, every rank
owns a slice of the global batch, and the only exchange is one all-reduce of the {satisfied, failed}
counters plus the max-over-ranks timing.  No line of the product runs here -- there is no GPU in this
tier, so the per-shard verdicts come from the oracle.  The product's own N > 1 path (bench.py launching
its ranks, one Engine per rank, the reduction of the device counters) is run by tests/test_multi_rank.py
in the GPU tier."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, per_rank, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import __graft_entry__
    __graft_entry__.load_package()
    import oracle_lib
    from helpers import oracle_lane
    from zkinterface_ir_amd import workloads

    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    wl = workloads.ArithLayered(W=32, D=5, n_instance0=4, n_out=2)
    lane_offset = rank * per_rank
    inst, wit = wl.inputs(per_rank, lane_offset)
    # the shard is a slice of the global batch
    g_inst, g_wit = wl.inputs(per_rank * world)
    assert np.array_equal(inst, g_inst[lane_offset:lane_offset + per_rank])
    assert np.array_equal(wit, g_wit[lane_offset:lane_offset + per_rank])
    probe = wl.relation_messages(with_epilogue=False, free_last=False)
    outs = np.zeros((per_rank, wl.n_out, wl.width), dtype=np.uint8)
    for lane in range(per_rank):
        iv = [int.from_bytes(inst[lane, k].tobytes(), 'little') for k in range(wl.n_instance0)]
        wv = [int.from_bytes(wit[lane, k].tobytes(), 'little') for k in range(wl.n_witness)]
        run = oracle_lane(wl.mod_le, iv, wv, probe, wl.width, trace=False)
        for t, wid in enumerate(wl.output_wire_ids()):
            outs[lane, t] = np.frombuffer(run.get(wid).to_bytes(wl.width, 'little'), dtype=np.uint8)
    n_bad = wl.set_expected_outputs(inst, outs, lane_offset)
    ok, _, _ = oracle_lib.eval_batch(b''.join(wl.relation_messages()), wl.mod_le, inst.tobytes(), wl.n_instance,
                                     wit.tobytes(), wl.n_witness, wl.width, per_rank, 2)
    counts = torch.tensor([sum(ok), per_rank - sum(ok)], dtype=torch.int64)
    assert counts[1].item() == n_bad
    dist.all_reduce(counts)  # the one collective of the path
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py: max-over-ranks timing
    if rank == 0:
        with open(os.path.join(out_dir, 'result.txt'), 'w') as f:
            f.write('%d %d %d %.3f' % (counts[0].item(), counts[1].item(),
                                       workloads.expected_satisfied(per_rank * world), t.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_count_allreduce(tmp_path):
    world, per_rank = 2, 100  # corrupted global lanes: 0, 97 (rank 0) and 194 (rank 1)
    mp.spawn(_rank_main, args=(world, _free_port(), per_rank, str(tmp_path)), nprocs=world, join=True)
    sat, failed, expected, tmax = open(tmp_path / 'result.txt').read().split()
    assert int(sat) == int(expected) == 197 and int(failed) == 3
    assert float(tmax) == pytest.approx(0.002)
