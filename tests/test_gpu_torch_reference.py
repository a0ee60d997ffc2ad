"""An independent statement of the per-Linear path in plain PyTorch on the GPU (upstream's algorithm, SURVEY.md A.2:
per-sample fp32 `H += x^T x`, torch.linalg.cholesky / cholesky_inverse / cholesky(upper), the column loop in Python)
against this backend on the same inputs.  Not the oracle (that is `oracle/`, bit-exact given the factor): this one
shares no code with it and uses torch's own factorisations and matmuls, so it checks the restatement itself --
scales equal to the bit, integer levels equal except where two fp32 factorisations / summation orders round apart
(the error feedback then moves the rest of that row: `tools/torch_gptq_ref.py` measures 5.7e-4 of the levels at
4096 x 4096 and 1.0e-2 at 4096 x 14336)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _torch_gptq(Wb, X, gs=128, bs=128, symmetric=True):
    dev = Wb.device
    R, K = Wb.shape
    S = X.shape[0]
    H = torch.zeros(K, K, device=dev)
    for n in range(S):
        x = (2.0 / (n + 1)) ** 0.5 * X[n].float()
        H *= n / (n + 1)
        H += x.t() @ x
    W = Wb.float().clone()
    Wg = W.reshape(R, K // gs, gs)
    mn, mx = Wg.amin(-1).clamp_max(0), Wg.amax(-1).clamp_min(0)
    eps = torch.finfo(torch.float32).eps
    if symmetric:
        scale = (torch.maximum(-mn, mx) / torch.tensor(7.5, device=dev)).clamp_min(eps)
        zp = torch.zeros_like(scale)
    else:
        scale = ((mx - mn) / torch.tensor(15.0, device=dev)).clamp_min(eps)
        zp = torch.clamp(torch.round(-8.0 - mn / scale), -8, 7)
    perm = torch.argsort(torch.diag(H), descending=True, stable=True)
    W, H = W[:, perm], H[perm][:, perm].clone()
    g_of = (torch.arange(K, device=dev) // gs)[perm]
    dead = torch.diag(H) == 0
    H[dead, dead] = 1
    W[:, dead] = 0
    H += torch.eye(K, device=dev) * (0.01 * torch.mean(torch.diag(H)))
    U = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
    L = torch.zeros_like(W)
    for i1 in range(0, K, bs):
        i2 = min(i1 + bs, K)
        W1, Err, U1 = W[:, i1:i2].clone(), torch.zeros(R, i2 - i1, device=dev), U[i1:i2, i1:i2]
        for i in range(i2 - i1):
            w, sc, z = W1[:, i], scale[:, g_of[i1 + i]], zp[:, g_of[i1 + i]]
            lev = torch.clamp(torch.round(w / sc + z), -8, 7)
            L[:, i1 + i] = lev
            err = (w - (lev - z) * sc) / U1[i, i]
            W1[:, i:] -= err.unsqueeze(1) * U1[i, i:].unsqueeze(0)
            Err[:, i] = err
        W[:, i2:] -= Err @ U[i1:i2, i2:]
    return scale, zp, L[:, torch.argsort(perm)].to(torch.int8)


@pytest.mark.parametrize("R,K,S,T,symmetric", [(96, 512, 8, 64, True), (64, 768, 6, 96, False), (128, 1024, 12, 128, True),
                                               (1024, 4096, 64, 384, True)])      # a production in_features
def test_plain_torch_statement_of_the_path_agrees(dev, R, K, S, T, symmetric):
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_shared
    from quantool_amd.engine.schemes import QuantArgs

    torch.manual_seed(R + K)
    X = torch.randn(S, T, K, device=dev)
    X[..., :5] *= 8                                   # a few loud channels, so that the activation ordering matters
    X = X.to(torch.bfloat16)
    Wb = (torch.randn(R, K, device=dev) * 0.02).to(torch.bfloat16)
    acc = HessianAccumulator(K, dev)
    acc.add(X)
    res = gptq_quantize_shared([Wb], acc, QuantArgs(symmetric=symmetric, actorder="static"))[0]
    scale, zp, levels = _torch_gptq(Wb, X, symmetric=symmetric)
    torch.cuda.synchronize()
    assert torch.equal(res.scale_f32, scale)
    assert torch.equal(res.zp_f32, zp)
    packed = res.weight_packed
    nib = torch.stack([(packed >> (4 * j)) & 0xF for j in range(8)], dim=-1).reshape(R, -1)[:, :K].to(torch.int8) - 8
    rate = float((nib != levels).float().mean())
    assert rate < 5e-3, rate
