"""Per-Linear GPTQ on the device: Hessian accumulation -> prepare -> factorise -> sweep -> pack.

Host-side counterpart of upstream's ``accumulate_hessian`` / ``quantize_weight`` (SURVEY.md
section 8a rows a7-a11, a14), which quantool reaches through
``src/quantool/methods/llm_compressor/base.py:161``.  All arithmetic runs in the HIP library
(``quantool_amd.hip.ops``); torch supplies buffers, the stream and index bookkeeping.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from ..hip import ops
from .schemes import QuantArgs


class HessianAccumulator:
    """Raw Gram sum G = sum_b X_b^T X_b (fp32, lower triangle) and the sample count n.

    Upstream keeps the running average ``H = H*n/(n+B) ...; H += (2/n) X^T X`` per Linear
    (a7); ``(2/n) * G`` is the same matrix and the factor is applied in ``hessian_prepare``.
    Linears that read the same input (q/k/v, gate/up) share one accumulator: their upstream
    Hessians are byte-identical, so one X^T X pass and one factorisation serve them all.

    The reference's calling pattern is one sample per batch (``base.py:161``; SURVEY A.1: batch
    size 1, T <= 384 tokens).  A Gram launch that short cannot fill the chip and would re-read and
    re-write the whole K x K fp32 ``G`` per sample, so ``add`` appends small batches to a device
    token buffer ``[stage_tokens, K]`` and launches ``qt_xtx_accumulate`` once per full buffer
    (and on ``flush``, which every reader of ``G`` goes through).  Batches of ``direct_tokens`` or
    more skip the buffer.
    """

    STAGE_BYTES = 1 << 30      # default token buffer: 1 GiB worth of rows, 4096..65536 tokens
    DIRECT_TOKENS = 16384      # a batch this long is worth its own launch

    def __init__(self, K: int, device, stage_tokens: Optional[int] = None, dtype: Optional[torch.dtype] = None):
        self.K = K
        self.dtype = dtype          # 16-bit dtype of the staged tokens; None: taken from the first batch
        self._G = torch.zeros((K, K), dtype=torch.float32, device=device)
        self.n = 0
        if stage_tokens is None:
            stage_tokens = max(4096, min(65536, self.STAGE_BYTES // (2 * K)))
        self.stage_tokens = int(stage_tokens) // 64 * 64
        self._stage: Optional[torch.Tensor] = None
        self._stage_sized = False   # the fp32 path halves stage_tokens exactly once
        self._fill = 0

    @property
    def G(self) -> torch.Tensor:
        """The Gram sum with every staged token folded in."""
        self.flush()
        return self._G

    @G.setter
    def G(self, value: torch.Tensor) -> None:
        self._G = value

    def add(self, X: torch.Tensor, num_samples: Optional[int] = None) -> None:
        """X: [B, T, K] or [T, K].  ``num_samples`` defaults to B (upstream's num_added)."""
        if num_samples is None:
            num_samples = X.shape[0] if X.dim() == 3 else 1
        X2 = X.reshape(-1, self.K)
        t = X2.shape[0]
        if X2.dtype not in (torch.bfloat16, torch.float16):
            # an fp32 checkpoint: upstream accumulates inp.float(), i.e. an fp32 Gram product -- staged and accumulated
            # in fp32 through the three-plane product (ops.xtx_accumulate_f32) when this accumulator is (or can still
            # become) an fp32 one.  Anything else rounds the batch to the accumulator's 16-bit dtype, and that is never
            # silent: also a wide batch arriving AFTER 16-bit ones goes through the policy (logged once, or refused).
            if ops.wide_gram_mode() == "exact" and self.dtype in (None, torch.float32):
                if X2.dtype != torch.float32:
                    X2 = X2.float()
                self.dtype = torch.float32
            else:
                ops.wide_activation_policy(X2.dtype)
        self.n += int(num_samples)      # after the policy: a refused batch is not counted
        if t == 0:
            return
        if self.dtype is None:
            # the checkpoint's own dtype (the reference injects none, base.py:222-241); see ops.as_act16
            self.dtype = X2.dtype if X2.dtype in (torch.bfloat16, torch.float16) else torch.bfloat16
        if self.dtype == torch.float32 and not self._stage_sized:
            # same bytes of staging at 4 B / element -- ONCE per accumulator (halving on every direct batch, or after
            # every release_stage(), would walk the buffer down to 128 rows and turn every short batch into its own
            # read-modify-write of G)
            self._stage_sized = True
            if self.stage_tokens > 0:
                self.stage_tokens = max(128, self.stage_tokens // 2 // 128 * 128)
        if self.stage_tokens <= 0 or t >= min(self.DIRECT_TOKENS, self.stage_tokens):
            self._gram(X2 if X2.dtype == self.dtype else X2.to(self.dtype))
            return
        if self._stage is None:
            self._stage = torch.empty((self.stage_tokens, self.K), dtype=self.dtype, device=self._G.device)
        if self._fill + t > self.stage_tokens:
            self.flush()
        self._stage[self._fill:self._fill + t].copy_(X2)     # also the dtype conversion, if any
        self._fill += t

    def _gram(self, rows: torch.Tensor) -> None:
        # QT_XTX_LAUNCH_TOKENS=n (0 = off, the default): at most n tokens per Gram launch.  The kernel adds all tokens of
        # a launch into ONE fp32 accumulator per tile element; upstream adds one 384-token product per sample into H.  Both
        # are fp32 sums of the same exact products, but the long chain's rounding error grows with sqrt(tokens): against
        # the fp64 Gram the default is 2.5e-6 of sqrt(Hii Hjj) at K = 14336 where upstream's order gives 7e-7 (DESIGN.md
        # 2.0).  Launches of n tokens make it a two-level sum (partial sums of n, then their sum), at the price of one
        # read-modify-write of G per launch (K = 14336: 0.2 ms each).
        limit = _launch_token_limit()
        if limit <= 0 or rows.shape[0] <= limit:
            chunks = [rows]
        else:
            chunks = [rows[t0:t0 + limit] for t0 in range(0, rows.shape[0], limit)]
        for part in chunks:
            if part.dtype == torch.float32:
                ops.xtx_accumulate_f32(part, self._G)
            else:
                ops.xtx_accumulate(part, self._G)

    def flush(self) -> None:
        if self._fill:
            self._gram(self._stage[:self._fill])
            self._fill = 0

    def release_stage(self) -> None:
        self.flush()
        self._stage = None

    def reset(self) -> None:
        self._fill = 0
        self._G.zero_()
        self.n = 0


def _launch_token_limit() -> int:
    import os

    try:
        return max(0, int(os.environ.get("QT_XTX_LAUNCH_TOKENS", "0") or 0)) // 64 * 64
    except ValueError:
        return 0


@dataclass
class GPTQResult:
    """Outputs for one Linear, named as the compressed-tensors state_dict names them (a14)."""
    weight_packed: Optional[torch.Tensor]       # int32 [R, ceil(K/8)] (4-bit) or None
    weight_q: Optional[torch.Tensor]            # int8 levels [R, K] when not 4-bit packed
    weight_scale: torch.Tensor                  # model dtype [R, G]
    weight_zero_point: Optional[torch.Tensor]   # int8 [R, G] (asymmetric only)
    weight_g_idx: Optional[torch.Tensor]        # int32 [K] (actorder == "group" only)
    weight_shape: torch.Tensor                  # int64 [2]
    loss: torch.Tensor                          # fp32 [R] per-row loss (upstream logs its sum)
    info: torch.Tensor                          # int32 [1]; != 0 -> Hessian not PD, U = I used
    scale_f32: torch.Tensor = field(repr=False, default=None)
    zp_f32: torch.Tensor = field(repr=False, default=None)
    Qt: torch.Tensor = field(repr=False, default=None)          # int8 [K, R] sweep order
    col_src: Optional[torch.Tensor] = field(repr=False, default=None)
    g_of_col: torch.Tensor = field(repr=False, default=None)    # int32 [K] group of original col

    def dequantized(self, dtype=torch.float32) -> torch.Tensor:
        """(q - zp) * scale in original column order -- what upstream writes back to the module."""
        if self.Qt is None:
            raise RuntimeError("the integer levels of this result were released (sequential driver, "
                               "QT_RESULT_DETAIL_BYTES): the module's weight holds the dequantised values")
        return ops.dequantize(self.Qt, self.scale_f32, self.zp_f32, self.g_of_col, self.col_src, dtype)


def _normalize_actorder(actorder) -> Optional[str]:
    if actorder is None or actorder is False:
        return None
    a = str(actorder).lower()
    if a in ("static", "weight"):
        return "static"
    if a == "group":
        return "group"
    if a in ("none", "false"):
        return None
    raise ValueError(f"unknown actorder {actorder!r} (expected None, 'static'/'weight' or 'group')")


def gptq_quantize_shared(weights: Sequence[torch.Tensor], acc: HessianAccumulator, qargs: QuantArgs, *,
                         block_size: int = 128, dampening_frac: float = 0.01,
                         scale_dtype: Optional[torch.dtype] = None,
                         keep: Optional[dict] = None) -> List[GPTQResult]:
    """Quantise every ``weights[i]`` ([R_i, K], bf16 / fp16 / fp32) that shares the input behind ``acc``.

    Follows upstream's ``quantize_weight`` order: observer, optional activation
    ordering, dead columns, damping, factorisation, block sweep; the rows of all weights are
    swept as one stacked matrix (rows are independent given U).
    """
    return gptq_quantize_batched([(weights, acc)], qargs, block_size=block_size, dampening_frac=dampening_frac,
                                 scale_dtype=scale_dtype, keeps=None if keep is None else [keep])[0]


def batch_chains_enabled() -> bool:
    """``QT_BATCH_CHAINS=0``: every Linear group through its own chain on its own stream (rounds 1-3; A/B runs and
    cross-checks -- the results are the same to the bit either way)."""
    import os

    return os.environ.get("QT_BATCH_CHAINS", "1") != "0"


def batchable(groups: Sequence[Tuple[Sequence[torch.Tensor], HessianAccumulator]]) -> List[List[int]]:
    """Indices of ``groups`` (Linear groups = weights sharing one accumulator) that can go through the chain together:
    equal in_features, at most ``ops.MAX_BATCH`` per batch, and -- because no 128-row tile of the stacked sweep may
    straddle two factors -- at most ONE group per batch whose row count is not a multiple of 128 (it goes last)."""
    by_k = {}
    for i, (ws, acc) in enumerate(groups):
        by_k.setdefault(acc.K, []).append(i)
    out = []
    for K, idx in by_k.items():
        aligned = [i for i in idx if sum(int(w.shape[0]) for w in groups[i][0]) % 128 == 0]
        ragged = [i for i in idx if i not in aligned]
        while aligned or ragged:
            take = aligned[:ops.MAX_BATCH - 1] if ragged else aligned[:ops.MAX_BATCH]
            aligned = aligned[len(take):]
            if ragged and len(take) < ops.MAX_BATCH:
                take.append(ragged.pop(0))
            out.append(take)
    return out


def gptq_quantize_batched(groups: Sequence[Tuple[Sequence[torch.Tensor], HessianAccumulator]], qargs: QuantArgs, *,
                          block_size: int = 128, dampening_frac: float = 0.01,
                          scale_dtype: Optional[torch.dtype] = None,
                          keeps: Optional[List[dict]] = None) -> List[List[GPTQResult]]:
    """``gptq_quantize_shared`` for several Linear groups of ONE in_features at once (``batchable`` says which): the
    groups' factorisations go through ``qt_cholesky_inverse_upper_batched`` (one chain of launches for all of them) and
    their rows through one stacked ``qt_gptq_sweep_grouped`` (rows know their group's factor).  Upstream quantises every
    Linear of a decoder layer inside one ``oneshot`` call (base.py:161); per group the results are BIT-IDENTICAL to
    ``gptq_quantize_shared`` (``tests/test_gpu_batched_chains.py``).  One group: the single-problem entry points."""
    n = len(groups)
    if n == 0:
        return []
    if int(block_size) != 128:
        # the sweep kernel keeps one 128-column block of U and W in LDS; a different block size changes
        # which updates are two-rounding rank-1 steps and which are the fma chain of the trailing
        # product, i.e. the bits -- refuse instead of silently using 128
        raise ValueError(f"block_size={block_size}: this backend implements upstream's default block_size=128 only")
    K = groups[0][1].K
    dev = groups[0][1].G.device
    for ws, acc in groups:
        if acc.n <= 0:
            raise ValueError("no calibration samples were accumulated for this Linear")
        if acc.K != K:
            raise ValueError(f"batched groups must share in_features ({acc.K} != {K})")
        for w in ws:
            if w.dim() != 2 or w.shape[1] != K:
                raise ValueError(f"weight shape {tuple(w.shape)} does not match in_features {K}")
    if n > ops.MAX_BATCH:
        raise ValueError(f"at most {ops.MAX_BATCH} groups per batch (see batchable())")
    actorder = _normalize_actorder(qargs.actorder)
    gs = qargs.kernel_group_size
    gsz = K if gs <= 0 else gs
    if K % gsz:
        raise ValueError(f"in_features {K} not divisible by group_size {gsz}")
    G = K // gsz
    rows = [[int(w.shape[0]) for w in ws] for ws, _ in groups]
    grp_rows = [sum(r) for r in rows]
    if any(r % 128 for r in grp_rows[:-1]):
        raise ValueError(f"stacked groups of {grp_rows} rows: every group but the last needs a multiple of 128 (batchable())")
    R = sum(grp_rows)
    row_end = [sum(grp_rows[:i + 1]) for i in range(n)]
    row_begin = [0] + row_end[:-1]
    ar = torch.arange(K, dtype=torch.int32, device=dev)
    g_orig = (ar // gsz).to(torch.int32)

    # ---- per group: activation ordering (a9), prepare (a8) into its slice of the batch -----------------
    # U's storage doubles as prepare's scratch (no K x K workspace kept per stream); +256 floats per problem
    ubuf = torch.empty((n, K * K + 256), dtype=torch.float32, device=dev)
    U = ubuf[:, :K * K].view(n, K, K)
    A = torch.empty((n, K, K), dtype=torch.float32, device=dev)
    perms, invs, deads = [], [], []
    for b, (ws, acc) in enumerate(groups):
        perm = inv = None   # perm: sweep position -> original column; inv: its inverse
        if actorder is not None:
            perm, inv = ops.argsort_desc(ops.hessian_diag(acc.G, acc.n))
        _, dead, _ = ops.hessian_prepare(acc.G, acc.n, dampening_frac, perm, A_out=A[b], scratch=ubuf[b])
        perms.append(perm)
        invs.append(inv)
        deads.append(dead)

    # ---- factorise all groups in one chain of launches (a8) -------------------------------------------
    if n == 1:
        _, info = ops.cholesky_inverse_upper(A[0], U_out=U[0])
    else:
        info = ops.cholesky_inverse_upper_batched(A, U)
    del A

    # ---- stacked fp32 working copy in sweep order, observer (a10) ------------------------------------
    Wf = torch.empty((R, K), dtype=torch.float32, device=dev)
    g_sweeps = []
    if actorder == "group":
        # qparams on the PERMUTED matrix (groups are runs of 128 sweep positions): gather first, then one observer
        # pass over the stacked rows
        for b, (ws, acc) in enumerate(groups):
            r0 = row_begin[b]
            for w, r in zip(ws, rows[b]):
                ops.weight_gather_f32(w if w.stride(1) == 1 else w.contiguous(), perms[b], deads[b], out=Wf[r0:r0 + r])
                r0 += r
            g_sweeps.append(g_orig)
        scale, zp, scale_t, zp_t = ops.group_minmax_qparams(Wf, gs, qargs.symmetric, qargs.num_bits)
    else:
        # qparams on the ORIGINAL matrix (before permutation / dead-column zeroing) and the working copy in sweep order
        # from ONE read of every weight; the group-major tables the sweep reads are written in place (a Linear's rows
        # are a column range of them)
        scale = torch.empty((R, G), dtype=torch.float32, device=dev)
        zp = torch.empty((R, G), dtype=torch.float32, device=dev)
        scale_t = torch.empty((G, R), dtype=torch.float32, device=dev)
        zp_t = torch.empty((G, R), dtype=torch.float32, device=dev)
        for b, (ws, acc) in enumerate(groups):
            r0 = row_begin[b]
            for w, r in zip(ws, rows[b]):
                w = w if w.stride(1) == 1 else w.contiguous()
                if K * w.element_size() <= 160 * 1024:          # a row fits the LDS: one pass
                    ops.weight_gather_qparams(w, perms[b], deads[b], gs, qargs.symmetric, qargs.num_bits, out=Wf[r0:r0 + r],
                                              scale=scale[r0:r0 + r], zp=zp[r0:r0 + r], scale_t=scale_t[:, r0:r0 + r],
                                              zp_t=zp_t[:, r0:r0 + r])
                else:                                           # (in_features beyond 81 920: the two passes of rounds 1-3)
                    ops.weight_gather_f32(w, perms[b], deads[b], out=Wf[r0:r0 + r])
                    s_, z_, st_, zt_ = ops.group_minmax_qparams(w, gs, qargs.symmetric, qargs.num_bits)
                    scale[r0:r0 + r], zp[r0:r0 + r] = s_, z_
                    scale_t[:, r0:r0 + r], zp_t[:, r0:r0 + r] = st_, zt_
                r0 += r
            g_sweeps.append(g_orig if perms[b] is None else g_orig[perms[b].long()].contiguous())

    # ---- the sweep (a11): all groups' rows at once, each row against its group's factor --------------
    if keeps is not None:  # stage boundaries for the parity tests
        for b in range(n):
            keeps[b].update(U=U[b].clone(), perm=perms[b], dead=deads[b])
    if n == 1:
        Qt, loss = ops.gptq_sweep(Wf, U[0], scale_t, zp_t, g_sweeps[0], block_size, qargs.num_bits)
    else:
        Qt, loss = ops.gptq_sweep_grouped(Wf, U, row_end, scale_t, zp_t, torch.stack(g_sweeps).contiguous(), block_size,
                                          qargs.num_bits)
    del Wf, U, ubuf

    # ---- outputs in original column order (a14) -------------------------------------------------------
    results: List[List[GPTQResult]] = []
    for b, (ws, acc) in enumerate(groups):
        col_src = invs[b]  # output column c lives at sweep position inv[c]
        if actorder == "group":
            g_of_col = g_sweeps[b][invs[b].long()].contiguous()     # upstream's saved weight_g_idx
        else:
            g_of_col = g_orig
        out: List[GPTQResult] = []
        r0 = row_begin[b]
        for w, r in zip(ws, rows[b]):
            Qt_i = Qt[:, r0:r0 + r].contiguous() if R > r else Qt
            sdt = scale_dtype or (w.dtype if w.dtype in (torch.bfloat16, torch.float16) else torch.float32)
            sc_i = scale[r0:r0 + r].contiguous()
            zp_i = zp[r0:r0 + r].contiguous()
            packed = ops.pack_int4(Qt_i, col_src) if qargs.num_bits == 4 else None
            wq = None
            if packed is None:
                wq = (Qt_i.t() if col_src is None else Qt_i[col_src.long()].t()).contiguous()
            out.append(GPTQResult(
                weight_packed=packed, weight_q=wq, weight_scale=sc_i.to(sdt),
                weight_zero_point=None if qargs.symmetric else zp_i.to(torch.int8),
                weight_g_idx=g_of_col if actorder == "group" else None,
                weight_shape=torch.tensor([r, K], dtype=torch.int64), loss=loss[r0:r0 + r], info=info[b:b + 1],
                scale_f32=sc_i, zp_f32=zp_i, Qt=Qt_i, col_src=col_src, g_of_col=g_of_col))
            r0 += r
        results.append(out)
    return results


def gptq_quantize_linear(weight: torch.Tensor, acc: HessianAccumulator, qargs: QuantArgs, **kw) -> GPTQResult:
    return gptq_quantize_shared([weight], acc, qargs, **kw)[0]
