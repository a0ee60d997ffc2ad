"""Edge cases the domain has (SURVEY 8c test list): empty and ragged inputs, maximum sizes
(70B's K = 28672 Hessian, > 65535 rows), argument errors through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from quantool_amd.hip import ops as _ops

    return _ops


def test_empty_token_batch_is_a_noop(ops, dev):
    G = torch.ones((64, 64), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(torch.empty((0, 64), dtype=torch.bfloat16, device=dev), G)
    s = torch.zeros(64, dtype=torch.float32, device=dev)
    ops.act_stats_accumulate(torch.empty((0, 64), dtype=torch.bfloat16, device=dev), abs_sum=s)
    torch.cuda.synchronize()
    assert bool((G == 1).all()) and bool((s == 0).all())


def test_no_calibration_samples_is_an_error(dev):
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_linear
    from quantool_amd.engine.schemes import QuantArgs

    acc = HessianAccumulator(128, dev)
    with pytest.raises(ValueError, match="no calibration samples"):
        gptq_quantize_linear(torch.zeros((8, 128), device=dev), acc, QuantArgs())


def test_argument_errors_come_back_as_status_codes(ops, dev):
    from quantool_amd.hip._lib import QT_ERR_INVALID, HipBackendError

    W = torch.zeros((8, 192), dtype=torch.float32, device=dev)
    with pytest.raises(ValueError):
        ops.group_minmax_qparams(W, 128)                      # 192 % 128 != 0 (host check)
    U = torch.eye(128, dtype=torch.float32, device=dev)
    st = torch.ones((1, 8), dtype=torch.float32, device=dev)
    g_idx = torch.zeros(128, dtype=torch.int32, device=dev)
    with pytest.raises(HipBackendError) as ei:
        ops.gptq_sweep(torch.zeros((8, 128), device=dev), U, st, st.clone(), g_idx, blocksize=64)
    assert ei.value.status == QT_ERR_INVALID and "blocksize" in str(ei.value)
    with pytest.raises(ValueError):
        ops.xtx_accumulate(torch.zeros((4, 64), dtype=torch.bfloat16), torch.zeros((64, 64)))   # CPU tensors


def test_more_than_65535_rows(ops, oracle, dev):
    """lm_head-sized row counts: row-indexed grids are chunked (gridDim.y <= 65535)."""
    R, K = 70000, 128
    g = torch.Generator(device=dev).manual_seed(0)
    W = (torch.randn((R, K), generator=g, device=dev) * 0.02).to(torch.bfloat16)
    scale, zp, st, zt = ops.group_minmax_qparams(W, 128, True, 4)
    Wf = ops.weight_gather_f32(W)
    torch.cuda.synchronize()
    want, _ = oracle.minmax_qparams(W.float().cpu().numpy(), 128, True, 4)
    np.testing.assert_array_equal(scale.cpu().numpy(), want)
    np.testing.assert_array_equal(st.cpu().numpy(), want.T)
    assert torch.equal(Wf, W.float())
    s = torch.rand(K, device=dev) + 0.5
    out = ops.scale_columns(W, s)
    assert torch.equal(out, (W.float() * s).to(torch.bfloat16))


def test_llama3_70b_down_proj_hessian_size(ops, dev):
    """K = 28672 (70B intermediate): 3.3 GB Hessian, 6328 tiles; residual check of the factor."""
    K, n = 28672, 4096
    g = torch.Generator(device=dev).manual_seed(1)
    X = torch.randn((n, K), generator=g, device=dev).to(torch.bfloat16)
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, G)
    i = torch.tensor([0, 5, 28671, 14336], device=dev)
    j = torch.tensor([0, 3, 28671, 100], device=dev)
    want = (X[:, i].double() * X[:, j].double()).sum(0)
    torch.testing.assert_close(G[i, j].double(), want, rtol=1e-5, atol=1e-3)
    A, dead, _ = ops.hessian_prepare(G, 64, 0.01, None)
    del G
    Hd_row0 = torch.flip(A[K - 1 - torch.arange(4, device=dev)][:, :], dims=(1,))   # rows 0..3 of Hd's lower part
    U, info = ops.cholesky_inverse_upper(A)
    torch.cuda.synchronize()
    assert int(info.item()) == 0
    assert bool(torch.isfinite(torch.diag(U)).all()) and bool((torch.diag(U) > 0).all())
    # (U^T U)[0, 0] * Hd-ish sanity: U[0,0]^2 = (Hd^-1)[0,0] >= 1 / Hd[0,0]
    assert float(U[0, 0]) ** 2 >= 1.0 / float(Hd_row0[0, 0]) * 0.999
    # a small stacked sweep against it (rows independent: 130 rows, ragged)
    W = (torch.randn((130, K), generator=g, device=dev) * 0.02)
    scale, zp, st, zt = ops.group_minmax_qparams(W, 128, True, 4)
    g_idx = (torch.arange(K, device=dev) // 128).to(torch.int32)
    Qt, loss = ops.gptq_sweep(W.clone(), U, st, zt, g_idx, 128, 4)
    torch.cuda.synchronize()
    assert int(Qt.min()) >= -8 and int(Qt.max()) <= 7 and bool(torch.isfinite(loss).all())


def test_entry_points_from_two_host_threads_on_two_streams(dev):
    """The C ABI is re-entrant per stream: two host threads, each with its own HIP stream, quantise
    different Linears at the same time and get the bytes a single-threaded run gets."""
    import threading

    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_linear
    from quantool_amd.engine.schemes import QuantArgs

    def job(seed, stream=None):
        g = torch.Generator(device="cpu").manual_seed(seed)
        K = 384 if seed % 2 else 512
        X = torch.randn((3, 200, K), generator=g).to(torch.bfloat16).to(dev)
        W = (torch.randn((72, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)
        torch.cuda.synchronize()

        def run():
            acc = HessianAccumulator(K, dev)
            acc.add(X)
            r = gptq_quantize_linear(W, acc, QuantArgs(actorder="static"))
            return r.weight_packed.clone(), r.weight_scale.clone()

        if stream is None:
            out = run()
        else:
            with torch.cuda.stream(stream):
                out = run()
            stream.synchronize()
        return out

    want = {seed: job(seed) for seed in (1, 2)}
    torch.cuda.synchronize()
    got, errors = {}, []

    def worker(seed):
        try:
            torch.cuda.set_device(dev)
            for _ in range(3):
                got[seed] = job(seed, torch.cuda.Stream(device=dev))
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(seed,)) for seed in (1, 2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    torch.cuda.synchronize()
    for seed in (1, 2):
        assert torch.equal(got[seed][0], want[seed][0]) and torch.equal(got[seed][1], want[seed][1])
