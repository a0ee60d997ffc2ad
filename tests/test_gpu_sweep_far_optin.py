"""QT_SWEEP_FAR=bf16x3 (opt-in): the sweep's far update as a three-plane bf16 product instead of the f32-MFMA fmaf
chain.  Not the parity contract -- the oracle fixes the chain -- so it is held to the default path by a RATE: the two
differ in ~1e-5 of the integer levels at production sizes (`tools/sweep_far_ab.py`, profiles/r03_sweep_far_ab.txt),
here on small shapes with ragged edges (rows and columns that are not multiples of the 256-wide tiles)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,K", [(256, 2048), (300, 1536), (1028, 4096)])
def test_bf16x3_far_update_stays_within_a_rate_of_the_chain(ops, dev, monkeypatch, R, K):
    torch.manual_seed(R + K)
    X = torch.randn(4 * K, K, device=dev)
    X[:, :7] *= 6
    X = X.to(torch.bfloat16)
    W = (torch.randn(R, K, device=dev) * 0.02).to(torch.bfloat16)
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, G)
    perm, inv = ops.argsort_desc(ops.hessian_diag(G, 8))
    A, dead, _ = ops.hessian_prepare(G, 8, 0.01, perm)
    U, info = ops.cholesky_inverse_upper(A)
    assert int(info.item()) == 0
    sc, zp, sct, zpt = ops.group_minmax_qparams(W, 128, True, 4)
    g_sweep = (torch.arange(K, device=dev, dtype=torch.int32) // 128)[perm.long()].contiguous()

    def sweep():
        Wf = ops.weight_gather_f32(W, perm, dead)
        Qt, loss = ops.gptq_sweep(Wf, U, sct, zpt, g_sweep, 128, 4)
        torch.cuda.synchronize()
        return Qt.clone(), loss.clone(), Wf

    monkeypatch.delenv("QT_SWEEP_FAR", raising=False)
    q0, l0, _ = sweep()
    q0b, _, _ = sweep()
    assert torch.equal(q0, q0b)                                   # the default is deterministic
    monkeypatch.setenv("QT_SWEEP_FAR", "bf16x3")
    q1, l1, _ = sweep()
    q1b, _, _ = sweep()
    assert torch.equal(q1, q1b)                                   # and so is the opt-in
    rate = float((q0 != q1).float().mean())
    assert rate < 2e-3, rate
    assert float((l0 - l1).abs().max() / l0.abs().max()) < 2e-2
    # the first batch of blocks has no far update behind it: those columns are the chain's own in both modes
    first = 128 * 4 if K < 8192 else 128 * 8
    assert torch.equal(q0[:first], q1[:first])
