"""The fp32 TN GEMM building block: modes, edges, split-K, the structural-zero skip, and the
property the parity contract rests on -- its accumulation IS an ascending-k fmaf chain."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(dev):
    from quantool_amd.hip import ops as _ops

    return _ops


@pytest.mark.parametrize("M,N,k", [(64, 64, 16), (128, 256, 128), (130, 70, 37), (1, 5, 3), (256, 640, 1000),
                                   (128, 4096, 2048)])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_sgemm_modes_and_edges(ops, dev, M, N, k, mode):
    g = torch.Generator(device=dev).manual_seed(M * N + k)
    A = torch.randn((k, M), generator=g, device=dev)
    B = torch.randn((k, N), generator=g, device=dev)
    C = torch.randn((M, N), generator=g, device=dev)
    out = ops.sgemm_tn(A, B, C if mode == 0 else None, mode)
    ref = A.double().t() @ B.double()
    want = {0: C.double() - ref, 1: ref, 2: -ref}[mode]
    torch.testing.assert_close(out.double(), want, rtol=0, atol=2e-6 * k ** 0.5 * 16)


def test_sgemm_is_an_ascending_k_fmaf_chain(ops, dev, oracle):
    """Bit-for-bit equality with fmaf(a_k, b_k, acc) for k = 0..127 from acc = 0 -- the order the
    oracle fixes for upstream's `Err1 @ Hinv[i1:i2, i2:]` (oracle/gptq_oracle.c:orc_gptq_sweep)."""
    rng = np.random.default_rng(0)
    k, M, N = 128, 96, 160
    A = rng.standard_normal((k, M)).astype(np.float32)
    B = rng.standard_normal((k, N)).astype(np.float32)
    out = ops.sgemm_tn(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), None, 1).cpu().numpy()
    lib = oracle.lib()
    # reuse the C oracle's sweep on a crafted problem: one block, errors = A columns ... simpler: do
    # the chain directly with libm fmaf through ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.fmaf.restype = ctypes.c_float
    libm.fmaf.argtypes = [ctypes.c_float] * 3
    idx = [(0, 0), (5, 17), (95, 159), (40, 3)]
    for (m, n) in idx:
        acc = 0.0
        for kk in range(k):
            acc = libm.fmaf(float(A[kk, m]), float(B[kk, n]), acc)
        assert np.float32(acc) == out[m, n]


def test_sgemm_split_k_and_zero_skip(ops, dev):
    g = torch.Generator(device=dev).manual_seed(1)
    k, M, N = 4096, 128, 1024
    A = torch.randn((k, M), generator=g, device=dev)
    B = torch.randn((k, N), generator=g, device=dev)
    ref = A.double().t() @ B.double()
    out1 = ops.sgemm_tn(A, B, None, 1, allow_split_k=True)
    out2 = ops.sgemm_tn(A, B, None, 1, allow_split_k=True)
    torch.testing.assert_close(out1.double(), ref, rtol=0, atol=2e-3)
    assert torch.equal(out1, out2)                      # split-K reduction is ordered: deterministic
    # B lower-triangular-like (B[k][n] = 0 for k < n): skipping those k must not change the result
    Bl = torch.tril(torch.randn((1024, 1024), generator=g, device=dev))
    Al = torch.randn((1024, 128), generator=g, device=dev)
    full = ops.sgemm_tn(Al, Bl, None, 1)
    skip = ops.sgemm_tn(Al, Bl, None, 1, skip_zero_k=True)
    assert torch.equal(full, skip)


def test_sgemm_sub_large_ragged_tiles_match_small_tile_path(ops, dev):
    """>= 384 tiles of 128x128 select the 128x128 MODE_SUB kernel (8 waves, four per SIMD); fewer select the
    64x64 kernel.  Per output element both run the same ascending-k fmaf chain and one subtraction, so the
    ragged 2000 x 3100 x 200 product must equal, bit for bit, the same product assembled from two column
    halves that each take the small-tile path -- and a few elements are checked against libm's fmaf."""
    g = torch.Generator(device=dev).manual_seed(11)
    k, M, N = 200, 2000, 3100
    A = torch.randn((k, M), generator=g, device=dev)
    B = torch.randn((k, N), generator=g, device=dev)
    C = torch.randn((M, N), generator=g, device=dev)
    big = ops.sgemm_tn(A, B, C, 0)
    h = N // 2
    left = ops.sgemm_tn(A, B[:, :h].contiguous(), C[:, :h].contiguous(), 0)
    right = ops.sgemm_tn(A, B[:, h:].contiguous(), C[:, h:].contiguous(), 0)
    torch.cuda.synchronize()
    assert torch.equal(big, torch.cat([left, right], 1))
    libm = ctypes.CDLL("libm.so.6")
    libm.fmaf.restype = ctypes.c_float
    libm.fmaf.argtypes = [ctypes.c_float] * 3
    An, Bn, Cn, out = A.cpu().numpy(), B.cpu().numpy(), C.cpu().numpy(), big.cpu().numpy()
    for (m, n) in [(0, 0), (1999, 3099), (1920, 3072), (127, 128), (1000, 1555)]:
        acc = 0.0
        for kk in range(k):
            acc = libm.fmaf(float(An[kk, m]), float(Bn[kk, n]), acc)
        assert np.float32(Cn[m, n]) - np.float32(acc) == out[m, n]


@pytest.mark.parametrize("M,N,k,min_tiles", [
    (128, 128, 64, 1),          # one tile, the shortest k the ring kernel takes (4 stages)
    (256, 384, 128, 1),         # 6 tiles on 6 workgroups
    (384, 128 * 9, 512, 1),     # 27 tiles
    (4096, 128 * 40, 128, 1),   # 1280 tiles on 512 persistent workgroups: 2-3 tiles per workgroup, k = 128 (8 stages / tile)
    (2048, 128 * 33, 64, 1),    # 528 tiles, 4 stages per tile: the C batches sit as close together as they can
    (1024, 128 * 70, 1024, 1),  # 560 tiles, long k
    (1024, 128 * 68, 256, 1),   # 544 tiles: 32 workgroups get two tiles, the rest one
    (3072, 128 * 44, 128, 1),   # 1056 tiles
])
def test_ring_kernel_is_bit_identical_to_the_register_staged_kernel(ops, dev, M, N, k, min_tiles, monkeypatch):
    """MODE_SUB through sgemm_ring_kernel (LDS-DMA ring, persistent over tiles, counted vmcnt) and through
    sgemm_tn_kernel: per output element the same ascending-k fmaf chain from 0 and one subtraction, so every bit must
    agree -- with C a column slice of a wider matrix (ldc > N), in place, as the sweep's far update calls it."""
    g = torch.Generator(device=dev).manual_seed(M + N + k)
    A = torch.randn((k, M), generator=g, device=dev)
    Bw = torch.randn((k, N + 256), generator=g, device=dev)
    B = Bw[:, 256:]
    W0 = torch.randn((M, N + 256), generator=g, device=dev)
    outs = {}
    for ring in ("0", "1"):
        monkeypatch.setenv("QT_SGEMM_RING", ring)
        monkeypatch.setenv("QT_SGEMM_RING_MIN_TILES", str(min_tiles))
        W = W0.clone()
        C = W[:, 256:]
        for _ in range(2):                       # twice in place: the second pass reads what the first wrote
            ops.sgemm_tn(A, B, C, 0, out=C)
        torch.cuda.synchronize()
        outs[ring] = W
    assert torch.equal(outs["0"], outs["1"])
    assert torch.equal(outs["1"][:, :256], W0[:, :256])          # nothing outside the slice was touched
    ref = W0[:, 256:].double() - 2 * (A.double().t() @ B.double())
    torch.testing.assert_close(outs["1"][:, 256:].double(), ref, rtol=0, atol=4e-6 * k ** 0.5 * 16)


def test_ring_kernel_chains_through_the_sweep(ops, dev, monkeypatch):
    """The far update's chained form (four 128-deep chains per pass over W) runs on the ring kernel inside
    qt_gptq_sweep: levels, losses and the updated W must equal the register-staged kernel's bit for bit."""
    from tests.util import synth_activations, synth_weight, bits_to_bf16_tensor

    K, R = 2048, 1024
    X = bits_to_bf16_tensor(synth_activations(4 * K, K, seed=5), dev)
    W = torch.from_numpy(synth_weight(R, K, seed=6)).to(dev).to(torch.bfloat16)
    G = torch.zeros((K, K), dtype=torch.float32, device=dev)
    ops.xtx_accumulate(X, G)
    perm, inv = ops.argsort_desc(ops.hessian_diag(G, 8))
    A, dead, _ = ops.hessian_prepare(G, 8, 0.01, perm)
    U, info = ops.cholesky_inverse_upper(A)
    sc, zp, sct, zpt = ops.group_minmax_qparams(W, 128, True, 4)
    g_sweep = (torch.arange(K, device=dev, dtype=torch.int32) // 128)[perm.long()].contiguous()
    outs = {}
    for ring in ("0", "1"):
        monkeypatch.setenv("QT_SGEMM_RING", ring)
        monkeypatch.setenv("QT_SGEMM_RING_MIN_TILES", "1")
        Wf = ops.weight_gather_f32(W, perm, dead)
        Qt, loss = ops.gptq_sweep(Wf, U, sct, zpt, g_sweep, 128, 4)
        torch.cuda.synchronize()
        outs[ring] = (Qt, loss, Wf)
    for a, b in zip(outs["0"], outs["1"]):
        assert torch.equal(a, b)
