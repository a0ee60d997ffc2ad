"""N>1 path on CPU: world-size-2 gloo run of the sharding + final gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from quantool_amd.engine.sharding import (allreduce_gram, gather_state_dict, group_cost, lpt_assign, my_units,
                                          row_slices)


def test_lpt_assignment_is_balanced_and_deterministic():
    # one Llama-3-8B layer: qkv, o, gate_up (K=4096) and down (K=14336)
    N = 196608
    costs = [group_cost(4096, N, 6144), group_cost(4096, N, 4096), group_cost(4096, N, 28672),
             group_cost(14336, N, 4096)]
    own = lpt_assign(costs, 2)
    assert own == lpt_assign(costs, 2)
    assert own[3] != own[0] or own[3] != own[1]      # the heavy down_proj group does not share with everything
    costs32 = costs * 32
    own8 = lpt_assign(costs32, 8)
    loads = [sum(c for c, r in zip(costs32, own8) if r == k) for k in range(8)]
    assert max(loads) / min(loads) < 1.05
    assert sorted(sum((my_units(costs32, 8, r) for r in range(8)), [])) == list(range(len(costs32)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        units = [f"layers.{i}.mlp.down_proj" for i in range(5)]
        costs = [1.0 + i for i in range(5)]
        mine = my_units(costs, world, rank)
        local = {}
        for i in mine:
            g = torch.Generator().manual_seed(i)
            local[f"{units[i]}.weight_packed"] = torch.randint(-2 ** 31, 2 ** 31 - 1, (8, 4), generator=g,
                                                               dtype=torch.int32)
            local[f"{units[i]}.weight_scale"] = torch.randn(8, 2, generator=g).to(torch.bfloat16)
        # partitioning B: token-split Gram partials summed across ranks
        g = torch.Generator().manual_seed(7)
        X = torch.randn(64, 300, generator=g, dtype=torch.float64)      # K = 300: two 256-row bands
        mine_only = X[rank::world].t() @ X[rank::world]
        part = mine_only.clone()
        n_tot = allreduce_gram(part, 32 // world + (1 if rank == 0 else 0))
        full = X.t() @ X
        assert n_tot == 33
        # the bands of the lower triangle (what the Gram kernel writes) are summed over the ranks ...
        assert torch.allclose(part[:256, :256], full[:256, :256]) and torch.allclose(part[256:], full[256:])
        # ... and nothing else travels: the block above the diagonal tiles keeps the local partial
        assert torch.equal(part[:256, 256:], mine_only[:256, 256:])
        merged = gather_state_dict(local, dst=0)
        if rank == 0:
            ok = len(merged) == 10
            for i in range(5):
                g = torch.Generator().manual_seed(i)
                want_p = torch.randint(-2 ** 31, 2 ** 31 - 1, (8, 4), generator=g, dtype=torch.int32)
                want_s = torch.randn(8, 2, generator=g).to(torch.bfloat16)
                ok &= torch.equal(merged[f"{units[i]}.weight_packed"], want_p)
                ok &= torch.equal(merged[f"{units[i]}.weight_scale"], want_s)
            q.put(bool(ok))
        else:
            assert merged is None
    finally:
        dist.destroy_process_group()


def test_gather_state_dict_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_row_slices_cover_every_row_once():
    for rows, world in [(10, 4), (2, 4), (4096, 8), (1, 2), (37, 2)]:
        sl = row_slices(rows, world)
        assert len(sl) == world and sl[0][0] == 0 and sl[-1][1] == rows
        assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        sizes = [e - b for b, e in sl]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
