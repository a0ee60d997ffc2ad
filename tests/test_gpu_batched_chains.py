"""Batched chains over independent Hessians of equal K: upstream quantises every Linear of a
decoder layer inside one ``oneshot`` call (``/root/reference/src/quantool/methods/llm_compressor/base.py:161`` ->
``quantize_weight`` per Linear); here the factorisations (and sweeps) of a layer's equal-width input groups go through
the chain TOGETHER -- one panel-kernel launch serves all problems -- and every problem's result must be bit-identical to
the single-problem path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _damped_flipped(K, n_tokens, dev, seed):
    from quantool_amd.hip import ops

    g = torch.Generator(device=dev).manual_seed(seed)
    X = torch.randn((n_tokens, K), generator=g, device=dev)
    X[:, :: 37] *= 6.0
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X.to(torch.bfloat16), G)
    A, dead, _ = ops.hessian_prepare(G, 8, 0.01, None)
    return A


@pytest.mark.parametrize("K,n,g3", [(512, 3, False), (1152, 2, False), (1000, 3, False), (4096, 3, False), (2048, 2, True)])
def test_batched_factorisation_equals_single_bit_for_bit(dev, monkeypatch, K, n, g3):
    """Also with the bf16x3 block-row products forced on at a small K (g3), a ragged K (1000 = 7 x 128 + 104) and a
    padded stride between the problems."""
    from quantool_amd.hip import ops

    if g3:
        monkeypatch.setenv("QT_CHOL_G3_MIN_CHUNKS", "2")
    As = [_damped_flipped(K, 2 * K + 64 * b, dev, seed=100 + b) for b in range(n)]
    singles = [ops.cholesky_inverse_upper(a.clone()) for a in As]
    pad = 256
    Abuf = torch.empty((n, K * K + pad), dtype=torch.float32, device=dev)
    Ubuf = torch.full((n, K * K + pad), float("nan"), dtype=torch.float32, device=dev)
    Ab = Abuf[:, :K * K].view(n, K, K)
    Ub = Ubuf[:, :K * K].view(n, K, K)
    for b in range(n):
        Ab[b].copy_(As[b])
    info = ops.cholesky_inverse_upper_batched(Ab, Ub)
    torch.cuda.synchronize()
    assert info.tolist() == [0] * n
    for b in range(n):
        U1, i1 = singles[b]
        assert int(i1.item()) == 0
        assert torch.equal(Ub[b], U1), f"problem {b}: batched factor differs from the single-problem one"
    assert bool(torch.isnan(Ubuf[:, K * K:]).all())          # nothing written between the problems


def test_batched_factorisation_reports_a_bad_pivot_per_problem(dev):
    from quantool_amd.hip import ops

    K, n = 384, 3
    As = [_damped_flipped(K, 2 * K, dev, seed=7 + b) for b in range(n)]
    As[1][200, 200] = -5.0                                   # problem 1 is not positive definite
    Ab = torch.stack(As).contiguous()
    Ub = torch.empty_like(Ab)
    want = [ops.cholesky_inverse_upper(a.clone()) for a in As]
    info = ops.cholesky_inverse_upper_batched(Ab, Ub)
    torch.cuda.synchronize()
    assert [int(w[1].item()) for w in want] == info.tolist() and info[1].item() != 0 and info[0].item() == 0
    assert torch.equal(Ub[1], torch.eye(K, device=dev))      # upstream's LinAlgError fallback, for that problem only
    for b in (0, 2):
        assert torch.equal(Ub[b], want[b][0])


def _group(dev, K, row_counts, n_tokens, seed, dtype=torch.bfloat16):
    from quantool_amd.engine.gptq_linear import HessianAccumulator

    g = torch.Generator(device=dev).manual_seed(seed)
    X = torch.randn((n_tokens, K), generator=g, device=dev)
    X[:, torch.randperm(K, generator=g, device=dev)[: max(1, K // 50)]] *= 8.0
    acc = HessianAccumulator(K, dev)
    acc.add(X.to(dtype).reshape(4, n_tokens // 4, K))
    ws = [(torch.randn((r, K), generator=g, device=dev) * 0.02).to(dtype) for r in row_counts]
    return ws, acc


def _same(a, b):
    return (a is None and b is None) or (a is not None and b is not None and torch.equal(a, b))


@pytest.mark.parametrize("actorder,symmetric,bits", [("static", True, 4), (None, True, 4), ("group", False, 4), ("static", True, 8)])
def test_batched_groups_equal_their_single_group_runs_bit_for_bit(dev, actorder, symmetric, bits):
    """Three Linear groups of one in_features (two Linears, one Linear, and a ragged 200-row one that must go last),
    different Hessians: factorised in one batched chain, swept as one stacked matrix -- every output of every Linear equal
    to what ``gptq_quantize_shared`` gives for its group alone."""
    from quantool_amd.engine.gptq_linear import batchable, gptq_quantize_batched, gptq_quantize_shared
    from quantool_amd.engine.schemes import QuantArgs

    K = 640
    qa = QuantArgs(num_bits=bits, symmetric=symmetric, group_size=128 if bits == 4 else None,
                   strategy="group" if bits == 4 else "channel", actorder=actorder)
    groups = [_group(dev, K, [200], 2048, seed=3), _group(dev, K, [128, 256], 1536, seed=1), _group(dev, K, [256], 4096, seed=2)]
    order = batchable(groups)
    assert order == [[1, 2, 0]]                       # the ragged group last
    batch = [groups[i] for i in order[0]]
    keeps = [{} for _ in batch]
    got = gptq_quantize_batched(batch, qa, keeps=keeps)
    torch.cuda.synchronize()
    for (ws, acc), res, kp in zip(batch, got, keeps):
        k1 = {}
        want = gptq_quantize_shared(ws, acc, qa, keep=k1)
        torch.cuda.synchronize()
        assert torch.equal(kp["U"], k1["U"]) and _same(kp["perm"], k1["perm"]) and torch.equal(kp["dead"], k1["dead"])
        assert len(res) == len(want) == len(ws)
        for r, w in zip(res, want):
            for f in ("weight_packed", "weight_q", "weight_scale", "weight_zero_point", "weight_g_idx", "loss", "Qt",
                      "scale_f32", "zp_f32", "info"):
                assert _same(getattr(r, f), getattr(w, f)), f
            assert torch.equal(r.dequantized(), w.dequantized()) and r.weight_shape.tolist() == w.weight_shape.tolist()
