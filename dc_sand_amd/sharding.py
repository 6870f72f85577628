"""Beam-axis sharding across the GPUs of one node.

Every coefficient is independent, so rank g of G owns a contiguous beam range
and produces the matching column slab ``[t][c][a][b_lo:b_hi]`` of the global
tensor; outputs stay local.  The only shared state is the delay table
``delay_vals[A][B]``: ONE broadcast from rank 0 per delay-model update
(``torch.distributed``: RCCL over xGMI on GPUs, gloo in the CPU tests), as raw
bytes.  The reference has no counterpart (single GPU, no collectives).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .parameters import BeamformerParameters, delay_vals_dtype


@dataclass(frozen=True)
class BeamShard:
    rank: int
    world: int
    beam_lo: int
    beam_hi: int

    @property
    def n_beams(self) -> int:
        return self.beam_hi - self.beam_lo


def beam_range(n_beams_total: int, world: int, rank: int) -> BeamShard:
    """Contiguous, balanced split: the first ``n % world`` ranks get one extra beam."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    if n_beams_total < world:
        raise ValueError("fewer beams than ranks")
    base, rem = divmod(n_beams_total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return BeamShard(rank, world, lo, hi)


def local_parameters(global_params: BeamformerParameters, shard: BeamShard) -> BeamformerParameters:
    return global_params.with_beams(shard.n_beams)


def slice_table(global_table: np.ndarray, global_params: BeamformerParameters, shard: BeamShard) -> np.ndarray:
    """Host-side equivalent of ``dcs_bf_set_delays_from_global``: the compact
    ``[A][B_local]`` table of a shard."""
    t = np.asarray(global_table, dtype=delay_vals_dtype).reshape(global_params.NR_STATIONS, global_params.NR_BEAMS)
    return np.ascontiguousarray(t[:, shard.beam_lo:shard.beam_hi]).ravel()


def broadcast_table(table_bytes, src: int = 0, group=None):
    """Broadcast the table (a ``torch.uint8`` tensor of A*B*16 bytes, CPU for
    gloo or CUDA for RCCL) from ``src`` in place; returns the tensor."""
    import torch.distributed as dist

    dist.broadcast(table_bytes, src=src, group=group)
    return table_bytes
