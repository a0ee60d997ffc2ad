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
