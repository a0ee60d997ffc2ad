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
        # counts beyond 12 bits travel as two halves; a count that does not fit is refused by EVERY rank, after the
        # collective (rank 1 alone holds it: raising there first would leave rank 0 waiting in the all-reduce)
        assert allreduce_gram(mine_only.clone(), 5000 if rank == 0 else 70001) == 75001
        try:
            allreduce_gram(mine_only.clone(), 7 if rank == 0 else 1 << 24)
            raise AssertionError("a sample count of 2^24 must be refused")
        except ValueError as e:
            assert "1 rank(s)" in str(e)

        def refuse(*a, **kw):
            raise AssertionError("gather_state_dict issued an object collective")

        dist.all_gather_object = dist.gather_object = dist.broadcast_object_list = refuse
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


def test_plan_groups_splits_what_exceeds_a_fair_share():
    from quantool_amd.engine.sharding import group_cost, plan_groups

    # a Llama-3-8B layer on 8 ranks: down_proj (K = 14336) is two thirds of the work -> split (B);
    # the K = 4096 groups are whole units (A) on distinct ranks
    N = 196608
    costs = [group_cost(14336, N, 4096), group_cost(4096, N, 6144), group_cost(4096, N, 4096), group_cost(4096, N, 28672)]
    plan = plan_groups(costs, 8)
    assert plan[0] == ("B", -1)
    owners = [o for kind, o in plan[1:] if kind == "A"]
    assert len(set(owners)) == len(owners)                    # no rank gets two whole groups while others idle
    assert plan_groups(costs, 1) == [("A", 0)] * 4
    # every rank computes the same plan
    assert plan_groups(costs, 8) == plan
    # Llama-3-70B's K = 28672 group is split at any world size > 1
    big = [group_cost(28672, N, 8192), group_cost(8192, N, 10240), group_cost(8192, N, 8192), group_cost(8192, N, 57344)]
    assert plan_groups(big, 2)[0] == ("B", -1) and plan_groups(big, 8)[0] == ("B", -1)


def _row_split_worker(rank, world, port, q, variant):
    """Control flow of partitioning B's gather step on CPU tensors: the per-rank sweep is replaced by a
    stand-in (the real one needs a GPU); what is checked is who sweeps which rows and how the rows of
    every tensor come back together -- in ONE collective per call -- including a rank that owns no row
    of a weight, an asymmetric scheme's zero points, 8-bit levels and actorder="group"'s g_idx."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import types

        import quantool_amd.engine.gptq_linear as gl
        import quantool_amd.engine.sharding as sh

        bits, symmetric, actorder, wdt = variant
        K, gs = 16, 8
        seen_rows = []
        perm = torch.arange(K, dtype=torch.int32).flip(0) // gs

        def levels(w):
            return (w.float() * 10).round().clamp(-100, 100).to(torch.int8)

        def fake_shared(weights, acc, qargs, **kw):
            out = []
            for w in weights:
                seen_rows.append(int(w.shape[0]))
                out.append(types.SimpleNamespace(
                    weight_packed=(w[:, :2].float() * 1000).to(torch.int32) if bits == 4 else None,
                    weight_q=None if bits == 4 else levels(w),
                    weight_scale=w[:, :2].clone(),
                    weight_zero_point=None if symmetric else levels(w)[:, 3:5].contiguous(),
                    weight_g_idx=perm.clone() if actorder == "group" else None,
                    dequantized=lambda dtype, w=w: (w * 2).to(dtype)))
            return out

        calls = {"all_gather": 0, "other": 0}
        real_all_gather = dist.all_gather

        def counting_all_gather(*a, **kw):
            calls["all_gather"] += 1
            return real_all_gather(*a, **kw)

        def refuse(*a, **kw):
            calls["other"] += 1
            raise AssertionError("row-split gather issued a second kind of collective")

        gl.gptq_quantize_shared = fake_shared
        dist.all_gather = counting_all_gather
        dist.all_gather_object = dist.broadcast_object_list = dist.broadcast = refuse
        g = torch.Generator().manual_seed(5)
        Ws = [torch.randn(r, K, generator=g).to(wdt) for r in (7, 1, 4)]   # 1 row: rank 1 owns nothing of it
        qa = types.SimpleNamespace(actorder=actorder, num_bits=bits, symmetric=symmetric, kernel_group_size=gs)
        res = sh.gptq_quantize_row_split(Ws, None, qa, with_dequantized=True)
        ok = seen_rows == ([4, 1, 2] if rank == 0 else [3, 2])
        ok &= calls == {"all_gather": 1, "other": 0}
        for w, r in zip(Ws, res):
            if bits == 4:
                ok &= torch.equal(r.weight_packed, (w[:, :2].float() * 1000).to(torch.int32)) and r.weight_q is None
            else:
                ok &= torch.equal(r.weight_q, levels(w)) and r.weight_packed is None
            ok &= torch.equal(r.weight_scale, w[:, :2]) and r.weight_scale.dtype == wdt
            ok &= (r.weight_zero_point is None) if symmetric else torch.equal(r.weight_zero_point, levels(w)[:, 3:5])
            ok &= (r.weight_g_idx is None) if actorder != "group" else torch.equal(r.weight_g_idx, perm)
            ok &= torch.equal(r.dequantized(wdt), w * 2)
            ok &= r.weight_shape.tolist() == [w.shape[0], K]
        # a result that does not look like the layout every rank assumed is refused, not mis-sliced -- and refused by
        # EVERY rank after the collective: only rank 0 sweeps the single row of this weight and can see the mismatch;
        # raising there before the all-gather would leave rank 1 waiting in it.  The flag travels in
        # the gathered buffer instead.
        qa_bad = types.SimpleNamespace(actorder=actorder, num_bits=bits, symmetric=symmetric, kernel_group_size=4)
        try:
            sh.gptq_quantize_row_split(Ws[1:2], None, qa_bad)
            ok = False
        except RuntimeError as e:
            ok &= "rank(s) [0]" in str(e) and "layout" in str(e)
            ok &= ("layout every rank assumes" in str(e)) == (rank == 0)      # the detail only where it was seen
        ok &= calls == {"all_gather": 2, "other": 0}
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("variant", [(4, True, "static", torch.float32), (4, False, "group", torch.bfloat16),
                                     (8, False, None, torch.float16)],
                         ids=["w4-sym-static-f32", "w4-asym-group-bf16", "w8-asym-f16"])
def test_row_split_gather_world2_gloo(variant):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_row_split_worker, args=(r, 2, port, q, variant)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(q.get(timeout=10) for _ in range(2))
    assert got == {0: True, 1: True}


def test_row_slices_cover_every_row_once():
    for rows, world in [(10, 4), (2, 4), (4096, 8), (1, 2), (37, 2)]:
        sl = row_slices(rows, world)
        assert len(sl) == world and sl[0][0] == 0 and sl[-1][1] == rows
        assert all(a[1] == b[0] for a, b in zip(sl, sl[1:]))
        sizes = [e - b for b, e in sl]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_flatten_mixed_dtypes_alignment_and_target_device():
    """`_flatten` packs tensors of any dtype at 16-byte aligned offsets into one byte buffer on ONE device
    (given, or inferred), so a dict like `result_tensors()` -- int32 words, bf16 scales, int8 zero-points,
    int64 shape -- survives the trip (the mixed host/GPU case runs in test_gpu_oneshot_dist.py)."""
    from quantool_amd.engine.sharding import _flatten, _unflatten

    g = torch.Generator().manual_seed(3)
    state = {"a.weight_packed": torch.randint(-2 ** 31, 2 ** 31 - 1, (5, 3), generator=g, dtype=torch.int32),
             "a.weight_scale": torch.randn(5, 1, generator=g).to(torch.bfloat16),
             "a.weight_zero_point": torch.randint(-8, 8, (5, 1), generator=g, dtype=torch.int8),
             "a.weight_shape": torch.tensor([5, 24], dtype=torch.int64),
             "a.weight_g_idx": torch.arange(24, dtype=torch.int32).t()}
    flat, meta = _flatten(state, "cpu")
    assert flat.dtype == torch.uint8 and all(off % 16 == 0 for _, _, _, off, _ in meta)
    back = _unflatten(flat, meta)
    assert set(back) == set(state)
    for k, v in state.items():
        assert back[k].dtype == v.dtype and torch.equal(back[k], v)
    flat0, meta0 = _flatten({})
    assert flat0.numel() == 0 and meta0 == []
