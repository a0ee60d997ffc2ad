"""Multi-GPU sharding of the per-Linear work (partitioning A).

Every Linear group's {X^T X, factorisation, sweep, pack} is independent given its activations, so
units are assigned to ranks up front (LPT-greedy on a K^2*(N+R) cost) and processed with no
data-path collective.  The only exchange is the gather of the packed state to rank 0 at the end
(RCCL on GPUs: backend "nccl"; gloo in the CPU tests).  One process per GPU.
"""
from __future__ import annotations

import io
from typing import Dict, List, Sequence, Tuple

import torch


def group_cost(K: int, n_tokens: int, rows: int) -> float:
    """Relative cost of one Linear group: X^T X (N*K^2) + sweep (R*K^2) (+ the 2/3 K^3 factor)."""
    return float(K) * K * (n_tokens + rows) + (2.0 / 3.0) * float(K) ** 3


def lpt_assign(costs: Sequence[float], world: int) -> List[int]:
    """Longest-processing-time-first greedy: returns the owning rank of every unit.
    Deterministic (ties by index), so every rank computes the same assignment independently."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    owner = [0] * len(costs)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += costs[i]
    return owner


def my_units(costs: Sequence[float], world: int, rank: int) -> List[int]:
    return [i for i, r in enumerate(lpt_assign(costs, world)) if r == rank]


def _flatten(state: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, list]:
    meta, chunks, off = [], [], 0
    for name in sorted(state):
        t = state[name].contiguous()
        nbytes = t.numel() * t.element_size()
        meta.append((name, str(t.dtype).replace("torch.", ""), tuple(t.shape), off, nbytes))
        chunks.append(t.reshape(-1).view(torch.uint8))
        off += nbytes
    dev = chunks[0].device if chunks else torch.device("cpu")
    flat = torch.cat(chunks) if chunks else torch.empty(0, dtype=torch.uint8, device=dev)
    return flat, meta


def _unflatten(flat: torch.Tensor, meta: list) -> Dict[str, torch.Tensor]:
    out = {}
    for name, dtype, shape, off, nbytes in meta:
        out[name] = flat[off:off + nbytes].view(getattr(torch, dtype)).reshape(shape)
    return out


def gather_state_dict(local: Dict[str, torch.Tensor], dst: int = 0, group=None, device=None):
    """Gather every rank's {name: tensor} to ``dst``.  Returns the merged dict on ``dst``, None
    elsewhere.  One flat byte buffer per rank: one large transfer per peer (xGMI links are
    point-to-point, so few large messages beat many small ones)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    flat, meta = _flatten(local)
    if device is not None:
        flat = flat.to(device)
    metas = [None] * world
    dist.all_gather_object(metas, (meta, int(flat.numel())), group=group)
    # Posted as ONE batch so that RCCL runs the transfers as a group: the receives on ``dst`` then
    # progress concurrently, one per xGMI link, instead of one after the other.
    if rank == dst:
        merged = dict(_unflatten(flat, meta))
        bufs = {r: torch.empty(metas[r][1], dtype=torch.uint8, device=flat.device)
                for r in range(world) if r != dst and metas[r][1] > 0}
        if bufs:
            for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, r, group) for r, buf in bufs.items()]):
                q.wait()
        for r, buf in bufs.items():
            merged.update(_unflatten(buf, metas[r][0]))
        return merged
    if flat.numel():
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, flat, dst, group)]):
            q.wait()
    return None


def allreduce_gram(G: torch.Tensor, n_samples: int, group=None):
    """Partitioning B: when the calibration TOKENS of one Linear group are split over
    ranks, every rank accumulates its own partial Gram sum and the partials are summed once
    (RCCL all-reduce of K^2 fp32: 64 MB at K = 4096, 822 MB at K = 14336) before the factorisation.
    Returns the global sample count.  The sum order of an all-reduce is fixed by the ring, not by
    this code: bit-identical results across world sizes are not guaranteed (H within 1e-5 is)."""
    import torch.distributed as dist

    dist.all_reduce(G, op=dist.ReduceOp.SUM, group=group)
    n = torch.tensor([int(n_samples)], dtype=torch.int64, device=G.device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    return int(n.item())
