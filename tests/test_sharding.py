"""Beam-axis sharding on the CPU: the partition, the one
broadcast (gloo, world_size 2) and the equivalence "shard output == column slab
of the global tensor", proved with the oracle.  The GPU side of the same logic
(device gather) is tests/test_gpu_parity.py::test_beam_shard_gather."""
import os

import numpy as np
import pytest


def test_beam_range_partitions_exactly():
    from dc_sand_amd.sharding import beam_range

    for n, w in ((1024, 8), (4096, 8), (10, 3), (7, 7), (1000, 6)):
        shards = [beam_range(n, w, r) for r in range(w)]
        assert shards[0].beam_lo == 0 and shards[-1].beam_hi == n
        for a, b in zip(shards, shards[1:]):
            assert a.beam_hi == b.beam_lo
        sizes = [s.n_beams for s in shards]
        assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n
    with pytest.raises(ValueError):
        beam_range(3, 4, 0)
    with pytest.raises(ValueError):
        beam_range(8, 2, 2)


def test_shard_is_column_slab_of_global_tensor(oracle):
    from conftest import rand_table
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.sharding import beam_range, local_parameters, slice_table

    gp = BeamformerParameters(NR_CHANNELS=40, NR_STATIONS=5, NR_BEAMS=22)
    table = rand_table(gp.n_pairs, seed=8)
    full = oracle.generate(oracle.params_from(gp), table, 3, 2)
    for world in (2, 4, 8):
        for r in range(world):
            sh = beam_range(gp.NR_BEAMS, world, r)
            lp = local_parameters(gp, sh)
            loc = oracle.generate(oracle.params_from(lp), slice_table(table, gp, sh), 3, 2)
            assert np.array_equal(loc.view(np.uint32), np.ascontiguousarray(full[:, :, :, sh.beam_lo:sh.beam_hi]).view(np.uint32))


def _worker(rank: int, world: int, port: int, ret):
    import torch
    import torch.distributed as dist

    from conftest import rand_table
    from dc_sand_amd import BeamformerParameters
    from dc_sand_amd.sharding import beam_range, broadcast_table, local_parameters, slice_table
    from oracle import bf_oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        gp = BeamformerParameters(NR_CHANNELS=16, NR_STATIONS=4, NR_BEAMS=10)
        truth = rand_table(gp.n_pairs, seed=21)
        # only rank 0 holds the delay model; the others start from zeros
        host = truth.copy() if rank == 0 else np.zeros_like(truth)
        buf = torch.from_numpy(host.view(np.uint8))
        broadcast_table(buf, src=0)
        got = buf.numpy().view(truth.dtype)
        assert np.array_equal(got.view(np.uint32), truth.view(np.uint32))
        sh = beam_range(gp.NR_BEAMS, world, rank)
        lp = local_parameters(gp, sh)
        loc = orc.generate(orc.params_from(lp), slice_table(got, gp, sh), 5, 1)
        # whole-job checksum = sum of the ranks' checksums (no data-path collective needed)
        ck = torch.tensor([orc.checksum_of(loc) % (1 << 62)], dtype=torch.int64)
        dist.all_reduce(ck)
        full = orc.generate(orc.params_from(gp), truth, 5, 1)
        assert int(ck.item()) == orc.checksum_of(full) % (1 << 62)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_broadcast_and_shard_gloo_world2(oracle):
    import torch.multiprocessing as mp

    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert dict(ret) == {0: 1, 1: 1}
