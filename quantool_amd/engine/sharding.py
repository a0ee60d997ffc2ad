"""Multi-GPU sharding of the per-Linear work.

Partitioning A (first half of this file): whole units per rank.  Partitioning B (second half,
``gptq_quantize_token_split``): one Linear group spread over the ranks, with the path's one real
exchange step -- the all-reduce of the partial Gram sums.

Every Linear group's {X^T X, factorisation, sweep, pack} is independent given its activations, so
units are assigned to ranks up front (LPT-greedy on a K^2*(N+R) cost) and processed with no
data-path collective.  The only exchange is the gather of the packed state to rank 0 at the end
(RCCL on GPUs: backend "nccl"; gloo in the CPU tests).  One process per GPU.

State of the evidence: the driver has had no 8-GPU node so far, so every collective below has run over gloo (CPU
tests, and two ranks sharing one GPU) and never over RCCL.  They are written so that no rank can leave a collective
the others enter: nothing rank-local is validated before a collective, error conditions travel inside the exchanged
buffers and are raised by every rank alike afterwards.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

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


def _flatten(state: Dict[str, torch.Tensor], device=None) -> Tuple[torch.Tensor, list]:
    """One byte buffer holding every tensor at a 16-byte aligned offset (a view of the buffer as the
    tensor's dtype needs its offset divisible by the element size).  The tensors may live on different
    devices (``result_tensors`` keeps ``weight_shape`` on the host next to packed words on the GPU): each
    is moved to ``device`` -- default: the first accelerator any of them is on, else the host -- before
    the one concatenation."""
    meta, chunks, off = [], [], 0
    if device is None:
        device = next((t.device for t in state.values() if t.device.type != "cpu"), torch.device("cpu"))
    dev = torch.device(device)
    for name in sorted(state):
        t = state[name].to(dev).contiguous()
        nbytes = t.numel() * t.element_size()
        meta.append((name, str(t.dtype).replace("torch.", ""), tuple(t.shape), off, nbytes))
        chunks.append(t.reshape(-1).view(torch.uint8))
        pad = (-nbytes) % 16
        if pad:
            chunks.append(torch.zeros(pad, dtype=torch.uint8, device=dev))
        off += nbytes + pad
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
    point-to-point, so few large messages beat many small ones).

    No object collective: a rank's buffer is its tensors (16-byte aligned) followed by their table of contents
    (names, dtypes, shapes, offsets as UTF-8 JSON), and the only thing exchanged beforehand is a fixed-layout
    ``int64[2]`` per rank (payload bytes, table bytes) in ONE tensor all-gather -- on RCCL a device collective like
    the transfers themselves, with no pickling and no host round trip through a store."""
    import json

    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    flat, meta = _flatten(local, device)
    toc = torch.frombuffer(bytearray(json.dumps(meta).encode("utf-8")), dtype=torch.uint8).to(flat.device)
    payload = int(flat.numel())
    sizes = _comm(torch.tensor([payload, int(toc.numel())], dtype=torch.int64, device=flat.device), group)
    all_sizes = [torch.empty_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = [tuple(int(v) for v in t.tolist()) for t in all_sizes]
    # Posted as ONE batch so that RCCL runs the transfers as a group: the receives on ``dst`` then
    # progress concurrently, one per xGMI link, instead of one after the other.
    if rank == dst:
        merged = dict(_unflatten(flat, meta))
        bufs = {r: torch.empty(sum(all_sizes[r]), dtype=torch.uint8, device=flat.device)
                for r in range(world) if r != dst and all_sizes[r][0] > 0}
        if bufs:
            for q in dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, r, group) for r, buf in bufs.items()]):
                q.wait()
        for r, buf in bufs.items():
            n_pay, n_toc = all_sizes[r]
            meta_r = [(n, d, tuple(sh), o, b) for n, d, sh, o, b in
                      json.loads(bytes(buf[n_pay:n_pay + n_toc].cpu().tolist()).decode("utf-8"))]
            merged.update(_unflatten(buf, meta_r))
        return merged
    if payload:
        for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, torch.cat([flat, toc]), dst, group)]):
            q.wait()
    return None


def _lower_block_rows(K: int, bs: int = 256):
    """(row slice, column count) of every 256-row band of the lower triangle, diagonal tile included --
    exactly the part of G the Gram kernel writes."""
    for r0 in range(0, K, bs):
        r1 = min(K, r0 + bs)
        yield slice(r0, r1), r1


def allreduce_gram(G: torch.Tensor, n_samples: int, group=None):
    """Partitioning B: when the calibration TOKENS of one Linear group are split over
    ranks, every rank accumulates its own partial Gram sum and the partials are summed once before
    the factorisation.  Only the lower triangle of G is valid (the Gram kernel computes lower
    256 x 256 tiles), so only its 256-row bands travel: packed into one flat buffer, ONE all-reduce
    of ~K^2/2 fp32 (34 MB at K = 4096, 415 MB at K = 14336 -- half the full matrix) that also carries the
    sample count, unpacked in place.
    Returns the global sample count.  The sum order of an all-reduce is fixed by the ring, not by
    this code: bit-identical results across world sizes are not guaranteed (H within 1e-5 is)."""
    import torch.distributed as dist

    K = G.shape[0]
    bands = list(_lower_block_rows(K))
    # The sample count rides at the end of the same buffer as two 12-bit halves (sums of world <= 4096 such halves
    # are exact in fp32) plus a "my count does not fit" flag.  Nothing is validated BEFORE the collective: the counts
    # differ per rank, and a rank that raised on its own value would leave the others waiting in the all-reduce until
    # the RCCL timeout.  Every rank sees the same reduced words and raises (or not) together afterwards.
    n_local = int(n_samples)
    fits = 0 <= n_local < (1 << 24)
    words = [float(n_local & 0xFFF), float(n_local >> 12), 0.0] if fits else [0.0, 0.0, 1.0]
    count = torch.tensor(words, dtype=G.dtype, device=G.device)
    flat = torch.cat([G[rows, :cols].reshape(-1) for rows, cols in bands] + [count])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for rows, cols in bands:
        cnt = (rows.stop - rows.start) * cols
        G[rows, :cols] = flat[off:off + cnt].view(rows.stop - rows.start, cols)
        off += cnt
    lo, hi, bad = (int(round(v)) for v in flat[off:off + 3].tolist())
    if bad:
        raise ValueError(f"{bad} rank(s) hold a sample count outside [0, 2^24): it does not travel next to the Gram bands")
    return (hi << 12) + lo


# ---- partitioning B: one Linear group spread over the ranks ------------------------
def _comm(t: torch.Tensor, group=None) -> torch.Tensor:
    """Tensor as the process group's backend wants it: device tensors for RCCL, host for gloo."""
    import torch.distributed as dist

    return t.cpu() if dist.get_backend(group) == "gloo" and t.is_cuda else t


def allreduce_inplace(t: torch.Tensor, op=None, group=None) -> None:
    """``dist.all_reduce`` on a device tensor whatever the backend (gloo: through the host)."""
    import torch.distributed as dist

    c = _comm(t, group)
    dist.all_reduce(c, op=op if op is not None else dist.ReduceOp.SUM, group=group)
    if c is not t:
        t.copy_(c)


def row_slices(rows: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced [begin, end) row ranges, one per rank (the first ``rows % world`` ranks
    take one extra row; a rank may get none when rows < world)."""
    base, extra = divmod(rows, world)
    out, r0 = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((r0, r0 + n))
        r0 += n
    return out


def dist_world(group=None) -> Tuple[int, int]:
    """(world size, rank) of the default process group, (1, 0) when torch.distributed is not in use."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def plan_groups(costs: Sequence[float], world: int) -> List[Tuple[str, int]]:
    """How a layer's Linear groups are spread over ``world`` ranks.  A group whose cost alone
    exceeds a fair share (total / world) is SPLIT over all ranks (partitioning B: tokens for the Gram
    sum, rows for the sweep) -- down_proj is two thirds of a Llama layer, 70B's K = 28672 group more;
    the rest are whole units assigned LPT-greedy (partitioning A) on top of the split groups' even load.
    Returns ("B", -1) or ("A", owner) per group; deterministic, so every rank computes the same plan."""
    if world <= 1:
        return [("A", 0)] * len(costs)
    share = sum(costs) / world
    split = [c > share for c in costs]
    rest = [i for i, sp in enumerate(split) if not sp]
    owners = lpt_assign([costs[i] for i in rest], world)
    plan: List[Tuple[str, int]] = [("B", -1)] * len(costs)
    for i, o in zip(rest, owners):
        plan[i] = ("A", o)
    return plan


def allreduce_accumulator(acc, group=None) -> None:
    """Sum a ``HessianAccumulator`` over the ranks in place (lower-triangle bands of G and n)."""
    g = _comm(acc.G, group)
    n_total = allreduce_gram(g, acc.n, group)
    if g is not acc.G:
        acc.G.copy_(g)
    acc.n = n_total


class GatheredResult:
    """Full-height outputs of one Linear whose rows were swept on several ranks: the state_dict
    entries of a ``GPTQResult`` plus, when asked for, the dequantised weight in the model dtype."""

    def __init__(self, parts: dict, w_dq: Optional[torch.Tensor]):
        self.weight_packed = parts.get("weight_packed")
        self.weight_q = parts.get("weight_q")
        self.weight_scale = parts["weight_scale"]
        self.weight_zero_point = parts.get("weight_zero_point")
        self.weight_g_idx = parts.get("weight_g_idx")
        self.weight_shape = parts["weight_shape"]
        self._w_dq = w_dq

    def dequantized(self, dtype=torch.float32) -> torch.Tensor:
        if self._w_dq is None:
            raise RuntimeError("dequantised rows were not gathered (gptq_quantize_row_split(with_dequantized=True))")
        return self._w_dq.to(dtype)


def _row_split_schema(weights, qargs, K: int, with_dequantized: bool):
    """Per weight, the row-split outputs every rank holds for its rows: [(key, columns, dtype)].  A function of
    the quantisation arguments and the weight's dtype alone, so that every rank -- also one that swept no
    rows of this weight -- lays out the gather buffer the same way without asking the others."""
    gs = qargs.kernel_group_size
    G = 1 if gs <= 0 else K // gs
    out = []
    for w in weights:
        cols = [("weight_packed", (K + 7) // 8, torch.int32)] if int(qargs.num_bits) == 4 else \
               [("weight_q", K, torch.int8)]
        cols.append(("weight_scale", G, w.dtype if w.dtype in (torch.bfloat16, torch.float16) else torch.float32))
        if not qargs.symmetric:
            cols.append(("weight_zero_point", G, torch.int8))
        if with_dequantized:
            cols.append(("@dequantized", K, w.dtype))
        out.append(cols)
    return out


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


def gptq_quantize_row_split(weights: Sequence[torch.Tensor], acc, qargs, *, group=None, block_size: int = 128,
                            dampening_frac: float = 0.01, with_dequantized: bool = False) -> List[GatheredResult]:
    """Steps 3-4 of partitioning B on an accumulator that already holds the GLOBAL Gram sum: every rank
    factorises the same Hessian (replicated: 2/3 K^3 flop is cheaper than moving the K^2 factor), sweeps
    only its contiguous slice of every weight's rows (rows are independent given U), and the packed
    rows, scales (and, for the sequential driver's write-back, the dequantised rows) of ALL the group's
    weights travel in ONE all-gather of a byte buffer whose layout every rank derives from the arguments
    (no per-tensor collectives, no object collectives: one message per peer per Linear group)."""
    import torch.distributed as dist

    from .gptq_linear import gptq_quantize_shared

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    K = int(weights[0].shape[1])
    dev = weights[0].device
    slices = [row_slices(int(w.shape[0]), world) for w in weights]
    mine = [sl[rank] for sl in slices]
    local = [w[b:e] for w, (b, e) in zip(weights, mine) if e > b]
    res = iter(gptq_quantize_shared(local, acc, qargs, block_size=block_size, dampening_frac=dampening_frac)
               if local else [])
    schema = _row_split_schema(weights, qargs, K, with_dequantized)
    with_gidx = str(qargs.actorder).lower() == "group"

    def rank_bytes(r: int) -> int:
        n = _pad16(4 * K) if with_gidx else 0
        for sl, cols in zip(slices, schema):
            rows = sl[r][1] - sl[r][0]
            n += sum(_pad16(rows * c * torch.empty(0, dtype=dt).element_size()) for _, c, dt in cols)
        return n

    # 16-byte header per rank: byte 0 = "my outputs do not have the layout every rank assumes".  A mismatch is not
    # raised before the collective (only ranks that swept rows can see it; the others would wait in the all-gather):
    # it travels in the buffer and every rank raises the same error afterwards.
    HDR = 16
    width = HDR + max(rank_bytes(r) for r in range(world))
    send = torch.zeros(width, dtype=torch.uint8, device=dev)
    off = HDR
    layout_error = None
    g_idx_local = None
    results = []
    for (b, e), cols in zip(mine, schema):
        r = next(res) if e > b else None
        results.append(r)
        for key, c, dt in cols:
            nbytes = (e - b) * c * torch.empty(0, dtype=dt).element_size()
            if r is not None:
                t = r.dequantized(dt) if key == "@dequantized" else getattr(r, key)
                if t is None or tuple(t.shape) != (e - b, c) or t.dtype != dt:
                    layout_error = layout_error or (f"{key} is {None if t is None else (tuple(t.shape), t.dtype)}, "
                                                    f"the layout every rank assumes says {((e - b, c), dt)}")
                else:
                    send[off:off + nbytes] = t.contiguous().reshape(-1).view(torch.uint8)
            off += _pad16(nbytes)
        if r is not None and with_gidx and g_idx_local is None:
            g_idx_local = r.weight_g_idx
    if with_gidx:
        if g_idx_local is not None:
            send[off:off + 4 * K] = g_idx_local.to(torch.int32).contiguous().view(torch.uint8)
        off += _pad16(4 * K)
    if layout_error is not None:
        send[0] = 1
    send = _comm(send, group)
    bufs = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(bufs, send, group=group)
    bufs = [buf.to(dev) for buf in bufs]
    failed = [r for r, flag in enumerate(torch.stack([buf[0] for buf in bufs]).tolist()) if flag != 0]     # one host read
    if failed:
        raise RuntimeError(f"row-split gather: rank(s) {failed} produced outputs whose layout differs from the one every "
                           f"rank derives from the arguments" + (f" (here: {layout_error})" if layout_error else ""))

    offs = [HDR] * world
    out = []
    for w, sl, cols in zip(weights, slices, schema):
        R = int(w.shape[0])
        parts, w_dq = {}, None
        for key, c, dt in cols:
            es = torch.empty(0, dtype=dt).element_size()
            pieces = []
            for r in range(world):
                rows = sl[r][1] - sl[r][0]
                nbytes = rows * c * es
                if rows:
                    pieces.append(bufs[r][offs[r]:offs[r] + nbytes].view(dt).reshape(rows, c))
                offs[r] += _pad16(nbytes)
            full = torch.cat(pieces, 0)
            if key == "@dequantized":
                w_dq = full
            else:
                parts[key] = full
        parts["weight_shape"] = torch.tensor([R, K], dtype=torch.int64)
        out.append(GatheredResult(parts, w_dq))
    if with_gidx:
        # the same on every rank that swept rows (one Hessian, one permutation): taken from the first that did
        src = next(r for r in range(world) if any(sl[r][1] > sl[r][0] for sl in slices))
        g_idx = bufs[src][offs[src]:offs[src] + 4 * K].view(torch.int32).clone()
        for o in out:
            o.weight_g_idx = g_idx
    return out


def gptq_quantize_token_split(weights: Sequence[torch.Tensor], local_batches, qargs, *, group=None,
                              block_size: int = 128, dampening_frac: float = 0.01,
                              num_local_samples: int = None):
    """Quantise the Linears that share one input when the calibration TOKENS of that input are
    spread over the ranks -- the shape that balances a decoder layer whose down_proj alone is two
    thirds of the work, and Llama-3-70B's K = 28672 group (partitioning B):

    1. every rank accumulates X^T X over its own batches                      (no communication)
    2. one all-reduce of the partial Gram sums (lower-triangle bands) and of the sample counts
       (the exchange step)
    3. every rank factorises the same Hessian and sweeps only ITS rows of every weight
    4. one all-gather of the packed rows and their scales                     (the gather step)

    Returns, on every rank, {"weight_packed": [R, K/8] int32 (or "weight_q" int8 for 8-bit),
    "weight_scale", optional "weight_zero_point" / "weight_g_idx"} per weight, full height.
    Results are bit-identical to a single-rank run on a Gram matrix summed in the same order
    (for two ranks: G0 + G1), which is what ``tests/test_gpu_token_split.py`` checks."""
    from .gptq_linear import HessianAccumulator

    K = int(weights[0].shape[1])
    dev = weights[0].device
    acc = HessianAccumulator(K, dev)
    for xb in local_batches:
        acc.add(xb.to(dev))
    if num_local_samples is not None:
        acc.n = int(num_local_samples)
    allreduce_accumulator(acc, group)
    res = gptq_quantize_row_split(weights, acc, qargs, group=group, block_size=block_size, dampening_frac=dampening_frac)
    out = []
    for r in res:
        parts = {k: getattr(r, k) for k in ("weight_packed", "weight_q", "weight_scale", "weight_zero_point",
                                            "weight_g_idx") if getattr(r, k) is not None}
        parts["weight_shape"] = r.weight_shape
        out.append(parts)
    return out
