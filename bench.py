#!/usr/bin/env python3
"""Headline benchmark: quantized weights/sec, GPTQ int4 g128, Llama-3-8B-shaped, 512 calibration
samples x 384 tokens, on N MI355X (BASELINE.json `metric`, configs[1]).

A "step" is the whole per-Linear hot path for ONE decoder layer (7 Linears, 218 103 808 weights):
Gram accumulation X^T X over all 196 608 calibration tokens for each of the layer's 4 distinct
inputs, Hessian prepare, Cholesky-inverse factor, the GPTQ block sweep (actorder=static, the
upstream default) and int4 pack.  `--steps 32` is the whole 8B model.  Inputs (synthetic weights
and activations) are resident in HBM before the timed region.

Other BASELINE.json configs, each a SEPARATELY LABELLED line (its own `metric` string, never the headline):
  --method awq                 configs[2]: Llama-3-8B AWQ int4 g128 (20-point scale search + RTN + pack)
  --model llama-3-70b          configs[3]: one Llama-3-70B-shaped decoder layer per step (hidden 8192, inter 28672:
                               the K = 28672 down_proj group on one GPU), GPTQ int4 g128
  --model mixtral-8x7b         configs[4]: one Mixtral-8x7B-shaped decoder layer per step: attention + 8 experts whose
                               tokens come from a seeded top-2 router (ragged token counts), SmoothQuant on the
                               attention inputs + GPTQ int4 g128 (W4A8 preset) on every Linear, experts included

Multi-GPU: decoder layers are independent units in this synthetic per-Linear mode, so every rank
quantizes its own layers (weak scaling, no data-path collective) and rank 0 gathers the packed
state over RCCL at the end of the timed region (north_star: "RCCL ... only to gather the final
quantized state_dict").  `python bench.py --gpus N` starts its own N ranks when no launcher did.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

N_SAMPLES = 512
SEQ_LEN = 384
REHEARSE = os.environ.get("QT_BENCH_REHEARSE_GLOO") == "1"   # see main(): N>1 control flow on a one-GPU box
# --gpus 1 with a ONE-rank RCCL group: the barriers, the warm-up and final gather and the max-over-ranks reduction of
# the N>1 path on the real backend (device tensors, RCCL dtypes) -- all a one-GPU box can show of it
ONE_RANK_RCCL = os.environ.get("QT_BENCH_ONE_RANK_RCCL") == "1"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0


def synth_activations(n_tokens: int, K: int, seed: int, device) -> torch.Tensor:
    """X ~ N(0,1) bf16 [n_tokens, K], 1 % of channels x10, built in chunks."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    X = torch.empty((n_tokens, K), dtype=torch.bfloat16, device=device)
    gain = torch.ones(K, dtype=torch.float32, device=device)
    n_out = max(1, K // 100)
    idx = torch.randperm(K, generator=g, device=device)[:n_out]
    gain[idx] = 10.0
    step = 16384
    for t0 in range(0, n_tokens, step):
        t1 = min(n_tokens, t0 + step)
        X[t0:t1] = (torch.randn((t1 - t0, K), generator=g, device=device) * gain).to(torch.bfloat16)
    return X


def synth_weight(R: int, K: int, seed: int, device) -> torch.Tensor:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return (torch.randn((R, K), generator=g, device=device) * 0.02).to(torch.bfloat16)


_STREAMS = {}
_ACCS = {}


def _accumulator(K, dev, lane, slot):
    """One accumulator (G + token staging buffer) per (lane, group), reused across steps like the
    per-stream workspaces: all work on it is ordered on that group's stream."""
    from quantool_amd.engine.gptq_linear import HessianAccumulator

    key = (dev.index, lane, slot, K)
    acc = _ACCS.get(key)
    if acc is None:
        acc = _ACCS[key] = HessianAccumulator(K, dev)
    else:
        acc.reset()
    return acc


def accumulate(acc, X, n_samples, per_sample):
    """Feed the calibration activations of one Linear group to its Hessian accumulator: either as
    ONE resident [N, K] batch (BASELINE.md 2.2's per-Linear microbenchmark) or the way the plugin
    path delivers them -- one sample of SEQ_LEN tokens per call (reference base.py:161, batch size 1),
    which `HessianAccumulator` stages into its device token buffer."""
    if not per_sample:
        acc.add(X, num_samples=n_samples)
        return
    T = X.shape[0] // n_samples
    for i in range(n_samples):
        acc.add(X[i * T:(i + 1) * T], num_samples=1)
    acc.flush()


def smooth_group(lins, weights, X, norm_vec, alpha=0.5):
    """SmoothQuant stage of a mapping (reference smoothquant.py:77-84): channel min/max of the
    producer's output over all tokens, s = range^a / max|W|^(1-a), balance weights * s, producer vector / s,
    and what the smoothed producer now emits: X / s.  Returns (weights by name, X / s)."""
    from quantool_amd.engine.smoothquant import ChannelMinMax, apply_smoothing, smoothquant_scales
    from quantool_amd.hip import ops

    K = X.shape[1]
    stats = ChannelMinMax(K, X.device)
    stats.add(X)
    ws = [weights[n] for n, _ in lins]
    s = smoothquant_scales(stats, ws, alpha)
    new_w, _ = apply_smoothing(s, ws, [norm_vec])
    return dict(zip((n for n, _ in lins), new_w)), ops.scale_columns(X, s, divide=True)


# QT_BENCH_SKIP=gram|chain: DIAGNOSTIC runs that leave one half of the step out (how long do the chains take with no Gram
# pass beside them, and the reverse); the printed line is labelled "diagnostic" and is not a measurement of the metric.
SKIP = os.environ.get("QT_BENCH_SKIP", "")
BATCH_CHAINS = os.environ.get("QT_BATCH_CHAINS", "1") != "0"
# QT_BENCH_SHAPE_SCALE=d: every width of the model's layer divided by d (multiples of 128) -- control-flow rehearsals of
# configs[3] / configs[4] with N ranks on one GPU (tests/test_gpu_bench_rehearsal.py); the line is labelled, not a result.
SHAPE_SCALE = int(os.environ.get("QT_BENCH_SHAPE_SCALE", "1") or 1)
_DIAG_G = {}


def quantize_layer(shape, weights, acts, qargs, n_samples, overlap=True, lane=0, per_sample=False, smooth=None):
    """One step: the hot path over one decoder layer.  Returns the packed outputs.

    The layer's Linear groups are independent, so each group's chain (factorise, sweep, pack) runs on
    its own HIP stream -- since round 4 one chain per SET of equal-width groups (one batched factorisation, one
    stacked sweep: gptq_quantize_batched) -- so the latency-bound chain of one set overlaps the MFMA-bound work of
    another.  The Gram passes share one further stream, smallest in_features first (below).  `lane` selects one of
    `--lanes` stream sets so that consecutive layers (independent units in this per-Linear mode, exactly as across
    GPUs) can be in flight at once."""
    from quantool_amd.engine.gptq_linear import HessianAccumulator, batchable, gptq_quantize_batched

    dev = next(iter(acts.values())).device
    outs = {}
    groups = sorted(shape.groups, key=lambda g: -g[1]) if overlap else list(shape.groups)
    main = torch.cuda.current_stream(dev)
    # Gram passes of the layer on ONE stream of their own, smallest in_features first; the chains (factorise, sweep,
    # pack) on a stream per group behind them.  The Gram kernel fills the chip by itself: four of them launched at
    # once only take turns on the CUs, while one after the other each runs at its stand-alone rate and the short
    # groups' chains still start within the first milliseconds (same box, 12 steps: 78.9 ms/step with the Gram
    # launches at 0.50 of the MFMA peak vs 79.5 ms at 0.34 with a Gram pass per group stream,
    # profiles/r03_stream_policy_sweep.txt).  QT_BENCH_XTX_STREAM: "lane" (default; one Gram stream per layer in
    # flight), "group" (rounds 1-2: the Gram pass on its group's stream), "shared" (one for both layers in flight),
    # "prio" (as lane, high stream priority: slower); QT_BENCH_XTX_ORDER=big issues the largest group first.
    # Per-sample calls are staged by many tiny copies, which one stream would put end to end for all four groups
    # (86.3 vs 83.3 ms/step): that mode keeps the Gram sums on the group streams, as engine/oneshot.py does.
    xmode = os.environ.get("QT_BENCH_XTX_STREAM", "group" if per_sample else "lane") if overlap else "group"
    sx = None
    if xmode in ("shared", "prio", "lane"):
        key = (dev.index, "xtx") if xmode == "shared" else (dev.index, "xtx", lane)
        if key not in _STREAMS:
            _STREAMS[key] = torch.cuda.Stream(device=dev, priority=-1 if xmode == "prio" else 0)
        sx = _STREAMS[key]
        sx.wait_stream(main)
    pending = []

    def run_chains(members):
        """factorise + sweep + pack for one batch of groups (usually one group, or all groups of one in_features)"""
        st = members[0][5]
        for gname, K, lins, wts, acc, st_g, ev in members:
            if ev is not None:
                st.wait_event(ev)
                for t in [acc.G] + ([wts[n] for n, _ in lins] if wts is not weights else []):
                    t.record_stream(st)      # allocated on the Gram stream, read on this one
            elif st_g is not st:
                st.wait_stream(st_g)
        with torch.cuda.stream(st):
            res = gptq_quantize_batched([([wts[n] for n, _ in lins], acc) for _, _, lins, wts, acc, _, _ in members], qargs)
            for (_, _, lins, _, _, _, _), rs in zip(members, res):
                for (lname, _), r in zip(lins, rs):
                    outs[f"{lname}.weight_packed"] = r.weight_packed
                    outs[f"{lname}.weight_scale"] = r.weight_scale

    slot_of = {g[0]: gi % 4 for gi, g in enumerate(groups)}      # largest in_features -> slot 0, as in rounds 1-2
    order = groups if (sx is None or os.environ.get("QT_BENCH_XTX_ORDER") == "big") else sorted(groups, key=lambda g: g[1])
    for gname, K, lins in order:      # a group's chain is issued right behind its Gram pass (the host never runs ahead
        if overlap:                   # of the GPU by a whole layer of per-sample Gram calls)
            # stream slots per layer: "all4" = one per group; "two" = the largest-K group alone, the rest
            # share one stream (diagnostic knob; default all4)
            slot = slot_of[gname] if os.environ.get("QT_BENCH_GROUPING", "all4") == "all4" else min(slot_of[gname], 1)
            if (dev.index, lane, slot) not in _STREAMS:
                # QT_BENCH_CHAIN_PRIO=1 (diagnostic): the chain streams at high priority
                prio = -1 if os.environ.get("QT_BENCH_CHAIN_PRIO") == "1" else 0
                _STREAMS[(dev.index, lane, slot)] = torch.cuda.Stream(device=dev, priority=prio)
            st = _STREAMS[(dev.index, lane, slot)]
            st.wait_stream(main)
        else:
            st = main
        with torch.cuda.stream(sx if sx is not None else st):
            X, wts = acts[gname], weights
            if smooth and gname in smooth:
                wts, X = smooth_group(lins, weights, X, smooth[gname])
            # on the Gram stream a fresh accumulator (read later on the group's stream: record_stream below)
            acc = HessianAccumulator(K, dev) if sx is not None else _accumulator(K, dev, lane, slot_of[gname])
            if SKIP == "gram" and gname in _DIAG_G:      # diagnostic: the chains alone, on a Gram sum formed once
                acc.G, acc.n = _DIAG_G[gname], n_samples
            else:
                accumulate(acc, X, n_samples, per_sample)
                if SKIP == "gram":
                    _DIAG_G[gname] = acc.G
        ev = None
        if sx is not None:
            ev = torch.cuda.Event()
            ev.record(sx)
        if SKIP == "chain":                              # diagnostic: the Gram passes alone
            continue
        pending.append((gname, K, lins, wts, acc, st, ev))
        if not BATCH_CHAINS or sx is None:
            run_chains([pending.pop()])
    # Batched chains (default): the layer's groups of equal in_features go through ONE chain of launches -- one batched
    # factorisation, one stacked sweep (gptq_quantize_batched) -- on the stream of the first of them, behind the Gram
    # passes of all of them.  QT_BATCH_CHAINS=0: a chain per group on its own stream (rounds 1-3).
    if pending:
        for bi, idx in enumerate(batchable([([w[n] for n, _ in l], a) for _, _, l, w, a, _, _ in pending])):
            members = [pending[i] for i in idx]
            if overlap:
                # a batch (also a batch of one) runs on the stream slot of its index, not of its first member: a Mixtral
                # layer's two batches -- ten groups of K = 4096, eight of K = 14336 -- would otherwise both land on
                # slot 0 and run one after the other
                if (dev.index, lane, bi % 4) not in _STREAMS:
                    _STREAMS[(dev.index, lane, bi % 4)] = torch.cuda.Stream(device=dev)
                st_b = _STREAMS[(dev.index, lane, bi % 4)]
                st_b.wait_stream(main)
                members = [m[:5] + (st_b,) + m[6:] for m in members]
            run_chains(members)
    return outs


def awq_layer(shape, weights, acts, qargs):
    """One step of the AWQ mode (BASELINE.json configs[2]): the 20-point per-channel scale search,
    apply, observer + round-to-nearest + int4 pack for the four mappings of one decoder layer."""
    from quantool_amd.engine.awq_linear import awq_quantize_groups

    outs = {}
    results = awq_quantize_groups([([weights[n] for n, _ in lins], [acts[gname]]) for gname, K, lins in shape.groups], qargs)
    for (gname, K, lins), res in zip(shape.groups, results):
        for (lname, _), r in zip(lins, res):
            outs[f"{lname}.weight_packed"] = r.weight_packed
            outs[f"{lname}.weight_scale"] = r.weight_scale
    return outs


def stage_split(shape, weights, acts, n_samples, smooth=None):
    """Per-stage device time of one step with NOTHING overlapped (one stream, stage after stage, torch events),
    outside the timed region: where a layer's time goes when the streams do not hide the latency-bound chains.
    The sum is therefore larger than `ms_per_step`.  Second result: the stages' achieved rates on their ALGORITHMIC
    work -- TFLOP/s for the Gram pass (N K (K+1)), the factorisation (2/3 K^3, fp32-equivalent) and the
    sweep (R K^2), GB/s for the HBM-bound single passes."""
    from quantool_amd.hip import ops

    def timed(fn, undo=None):
        # once untimed first: a stage's first call at a new size pays for its buffers (multi-GB hipMallocs at
        # K = 28672 are hundreds of ms of host time with the GPU idle between the two events)
        warm = fn()
        del warm
        if undo is not None:
            undo()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        torch.cuda.synchronize()
        return out, e0.elapsed_time(e1)

    tot, work = {}, {}
    for gname, K, lins in shape.groups:
        X = acts[gname]
        wts = weights
        row = {}
        if smooth and gname in smooth:
            (wts, X), row["smooth"] = timed(lambda: smooth_group(lins, weights, X, smooth[gname]))
        Ws = [wts[n] for n, _ in lins]
        W = torch.cat(Ws, 0) if len(Ws) > 1 else Ws[0]
        G = torch.zeros((K, K), dtype=torch.float32, device=X.device)
        _, row["gram"] = timed(lambda: ops.xtx_accumulate(X, G), undo=G.zero_)
        diag = ops.hessian_diag(G, n_samples)
        (perm, inv), row["order"] = timed(lambda: ops.argsort_desc(diag))
        (A, dead, _), row["prepare"] = timed(lambda: ops.hessian_prepare(G, n_samples, 0.01, perm))
        (U, info), row["factor"] = timed(lambda: ops.cholesky_inverse_upper(A))
        # observer + working copy in ONE read of W (round 4: qt_weight_gather_qparams; rounds 1-3 timed two passes,
        # "gather" and "qparams")
        R_, G_ = int(W.shape[0]), K // 128
        Wf = torch.empty((R_, K), dtype=torch.float32, device=X.device)
        sc, zp = torch.empty((R_, G_), device=X.device), torch.empty((R_, G_), device=X.device)
        sct, zpt = torch.empty((G_, R_), device=X.device), torch.empty((G_, R_), device=X.device)
        _, row["gather_qparams"] = timed(lambda: ops.weight_gather_qparams(W, perm, dead, 128, True, 4, out=Wf, scale=sc, zp=zp,
                                                                          scale_t=sct, zp_t=zpt))
        g_sweep = (torch.arange(K, device=X.device, dtype=torch.int32) // 128)[perm.long()].contiguous()
        (Qt, loss), row["sweep"] = timed(lambda: ops.gptq_sweep(Wf, U, sct, zpt, g_sweep, 128, 4))
        _, row["pack"] = timed(lambda: ops.pack_int4(Qt, inv))
        for k, v in row.items():
            tot[k] = tot.get(k, 0.0) + v
        # ALGORITHMIC work of the stages: flops for the MFMA-bound ones, bytes for the single passes
        N, R = int(X.shape[0]), int(W.shape[0])
        for k, w in (("gram", N * K * (K + 1)), ("factor", 2 * K ** 3 // 3), ("sweep", R * K * K),
                     ("prepare", 4 * K * K), ("gather_qparams", 6 * R * K), ("pack", 3 * R * K // 2)):
            work[k] = work.get(k, 0) + w
        del G, A, U, Wf, Qt, W
    rates = {}
    for k, w in work.items():
        if tot.get(k, 0.0) > 0:
            flops = k in ("gram", "factor", "sweep")
            rates[k + ("_tflops" if flops else "_gbs")] = round(w / (tot[k] * 1e-3) / (1e12 if flops else 1e9), 1)
    return {k: round(v, 3) for k, v in tot.items()}, rates


def join_streams(dev):
    main = torch.cuda.current_stream(dev)
    for key, st in list(_STREAMS.items()):
        if key[0] == dev.index:
            main.wait_stream(st)


def host_cores() -> int:
    """Threads this process may actually use: affinity, capped by the cgroup CPU quota and by the
    16-core share a one-GPU box grants (more threads than cores only thrash)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("QT_CPU_THREADS", "16"))))


def cpu_baseline_port():
    """The oracle (kind "port") on this box's host cores, on a bounded sample of the same
    workload: q_proj (4096 x 4096) -- Hessian accumulation in upstream's own form (per-sample
    fp32 `H += x.T @ x`) for 64 of the 512 samples (extrapolated x8, stated in `sample`), then
    the LAPACK three-step inverse and the C sweep (OpenMP over rows) in full."""
    import numpy as np

    from oracle import reference_path as rp

    rp.build()
    cores = host_cores()
    torch.set_num_threads(cores)
    rp.set_num_threads(cores)
    R = K = 4096
    n_meas = 64
    rng = np.random.default_rng(2)
    W = (rng.standard_normal((R, K)) * 0.02).astype(np.float32)
    H = torch.zeros((K, K), dtype=torch.float32)
    xs = [torch.from_numpy(rng.standard_normal((SEQ_LEN, K)).astype(np.float32)) for _ in range(8)]
    t0 = time.perf_counter()
    n = 0
    for b in range(n_meas):
        x = xs[b % 8]
        H *= n / (n + 1)
        n += 1
        x = (2.0 / n) ** 0.5 * x
        H += x.t() @ x
    t_hess = (time.perf_counter() - t0) * (N_SAMPLES / n_meas)
    Hn = H.numpy()
    t0 = time.perf_counter()
    out = rp.quantize_weight(W, Hn, actorder="static", inverse="lapack")
    rp.pack_int4(out["q"])
    t_rest = time.perf_counter() - t0
    total = t_hess + t_rest
    return {
        "value": R * K / total, "unit": "weights/s", "cores": cores, "kind": "port",
        "sample": (f"q_proj 4096x4096 W4A16 g128 actorder=static: per-sample fp32 H+=x^T x timed on {n_meas} "
                   f"of {N_SAMPLES} samples x {SEQ_LEN} tokens and scaled x{N_SAMPLES // n_meas} "
                   f"({t_hess:.1f} s), + LAPACK inverse, C sweep, pack in full ({t_rest:.1f} s)"),
    }


def visible_gpu_count() -> int:
    """GPUs this process could open, counted WITHOUT loading the HIP runtime (the launcher parent must stay a process
    that never touched the GPU): KFD topology nodes that have SIMDs, capped by the render nodes this user can open
    (a container is usually handed a subset of the host's) and by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES /
    CUDA_VISIBLE_DEVICES.  -1 when the KFD sysfs tree is absent (the caller then asks torch, whose
    `device_count()` goes through `hipGetDeviceCount`, i.e. loads the runtime but creates no context)."""
    import glob
    import re

    props = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not props:
        return -1
    n = 0
    for p in props:
        try:
            m = re.search(r"^simd_count\s+(\d+)", Path(p).read_text(), re.M)
        except OSError:
            continue
        n += bool(m and int(m.group(1)) > 0)
    render = [d for d in glob.glob("/dev/dri/renderD*") if os.access(d, os.R_OK | os.W_OK)]
    if render:
        n = min(n, len(render))
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip()]))
    return n


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: start N ranks of this same command as
    child processes, one per device (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment,
    rendezvous on 127.0.0.1), and return the worst exit code.  The parent never touches the GPU (devices are
    counted from sysfs, `visible_gpu_count`) and never re-execs itself.  Refuses when fewer than N
    GPUs are visible instead of silently running a smaller job; a rank whose device does not exist still
    fails by itself at `set_device`, and takes the others down with it."""
    import socket
    import subprocess

    if not REHEARSE:
        have = visible_gpu_count()
        if have < 0:
            have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible; refusing to run a smaller job "
                  f"(QT_BENCH_REHEARSE_GLOO=1 rehearses the N-rank control flow on one GPU)", file=sys.stderr)
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), QT_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = rc or code
                    for q in pending:       # a rank died: the others would wait in a collective for ever
                        procs[q].terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)       # one step per layer in flight: the allocator's pools are sized
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-split", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run the layer's groups on one stream")
    ap.add_argument("--lanes", type=int, default=0,
                    help="layers in flight (independent stream sets); 0 = 3 for the Llama-3-8B layer, 2 for the larger "
                         "ones (round 4, batched chains: 8B 79.0 / 74.6 / 75.6 ms per step at 2 / 3 / 4 lanes on one box; "
                         "a third Mixtral layer in flight does not fit next to its 27 GB of batched workspaces)")
    ap.add_argument("--samples", type=int, default=N_SAMPLES, help=argparse.SUPPRESS)
    ap.add_argument("--launch-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--model", choices=["llama-3-8b", "llama-3-70b", "mixtral-8x7b"], default="llama-3-8b",
                    help="llama-3-8b: the headline (BASELINE.json configs[1]); llama-3-70b: configs[3], one 70B-shaped decoder "
                         "layer per step; mixtral-8x7b: configs[4], one Mixtral-shaped layer per step (seeded top-2 routing, "
                         "SmoothQuant + GPTQ).  Anything but the default prints its own, separately labelled metric")
    ap.add_argument("--method", choices=["gptq", "awq"], default="gptq",
                    help="gptq: the headline metric (BASELINE.json configs[1]); awq: configs[2], a second, separately "
                         "labelled line (20-point scale search + RTN + pack per decoder layer)")
    ap.add_argument("--accumulate", choices=["single", "per-sample"], default="single",
                    help="single: one resident [N, K] activation batch per Linear group (BASELINE.md 2.2); per-sample: "
                         "512 calls of 384 tokens per group, the plugin path's calling pattern (staged on the device)")
    ap.add_argument("--actorder", choices=["static", "group", "none"], default="static",
                    help="activation ordering (default: upstream's default, static); other values are diagnostics")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started without a launcher: become the launcher (nothing has touched the GPU yet)
        raise SystemExit(launch_ranks(args.gpus))

    if args.lanes <= 0:
        args.lanes = 3 if args.model == "llama-3-8b" else 2
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}: refusing to run a job of another size")
    if args.launch_probe:
        # CPU-side check of the launch path (tests/test_bench_launch.py): rendezvous over gloo, one
        # all-reduce of the ranks, no GPU work, no benchmark line
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist_mod.init_process_group(backend="gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        if world > 1:
            dist_mod.all_reduce(t)
            dist_mod.destroy_process_group()
        if rank == 0:
            print(json.dumps({"probe": True, "n_gpus": world, "rank_sum": int(t.item()),
                              "self_launched": os.environ.get("QT_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
        return
    dist = None
    if world > 1 or ONE_RANK_RCCL:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if REHEARSE:
            # one-GPU rehearsal of the N>1 control flow: every rank on cuda:0, gloo for the host side
            # (RCCL refuses two ranks on one device).  Numbers from this mode are not benchmark results.
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            local_rank = 0
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from quantool_amd.engine.model_shapes import MODEL_SHAPES
    from quantool_amd.engine.schemes import QuantArgs
    from quantool_amd.hip import _lib

    lib = _lib.load()  # raises if the HIP library is missing: there is no fallback
    shape = MODEL_SHAPES[args.model]
    if SHAPE_SCALE > 1:      # rehearsal of a big configuration's control flow at 1 / SHAPE_SCALE of its widths
        from quantool_amd.engine.model_shapes import scaled

        shape = scaled(shape, SHAPE_SCALE)
    if args.method == "awq" and args.model != "llama-3-8b":
        raise SystemExit("--method awq is BASELINE.json configs[2]: Llama-3-8B only")
    n_tokens = args.samples * SEQ_LEN
    qargs = QuantArgs(num_bits=4, symmetric=True, group_size=128,
                      actorder=None if args.actorder == "none" else args.actorder)

    weights, acts, smooth = {}, {}, None
    route_counts = None
    if args.model == "mixtral-8x7b":
        # tokens reach an expert through a seeded top-2 router: ~N/4 tokens per expert, ragged
        g = torch.Generator(device=dev)
        g.manual_seed(7 + 1000 * rank)
        top2 = torch.randn((n_tokens, 8), generator=g, device=dev).topk(2, dim=1).indices
        route = [torch.nonzero((top2 == e).any(dim=1)).flatten() for e in range(8)]
        route_counts = [int(r.numel()) for r in route]
        x_moe = synth_activations(n_tokens, shape.groups[0][1], seed=5 + 1000 * rank, device=dev)
    for gi, (gname, K, lins) in enumerate(shape.groups):
        if gname.startswith("expert"):
            e = int(gname[len("expert"):].split("_")[0])
            if gname.endswith("_in"):
                acts[gname] = x_moe[route[e]].contiguous()      # the expert's routed rows of the MoE input
            else:
                acts[gname] = synth_activations(route_counts[e], K, seed=2 + 17 * gi + 1000 * rank, device=dev)
        else:
            acts[gname] = synth_activations(n_tokens, K, seed=2 + 17 * gi + 1000 * rank, device=dev)
        for li, (lname, R) in enumerate(lins):
            weights[lname] = synth_weight(R, K, seed=100 * gi + li + 1000 * rank, device=dev)
    if args.model == "mixtral-8x7b":
        del x_moe
        # SmoothQuant stage: q/k/v <- input_layernorm (upstream's Mixtral mapping smooths attention and the router
        # only; experts are quantised un-smoothed)
        g = torch.Generator(device=dev)
        g.manual_seed(11)
        smooth = {"attn_in": (1.0 + 0.1 * torch.randn(shape.groups[0][1], generator=g, device=dev)).to(torch.bfloat16)}
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    per_sample = args.accumulate == "per-sample"
    awq = args.method == "awq"

    def step(i):
        if awq:
            return awq_layer(shape, weights, acts, qargs)
        return quantize_layer(shape, weights, acts, qargs, args.samples, overlap=not args.no_overlap,
                              lane=i % max(1, args.lanes), per_sample=per_sample, smooth=smooth)

    if not awq:
        # set-up, before the W warm-up steps: one step per layer in flight, so that every stream set has its workspaces and
        # the caching allocator its pools (a pool that grows inside the timed region is a hipMalloc on the host's critical
        # path: `allocator_calls_in_timed_region` in the line says whether any was left)
        for lane_i in range(max(1, args.lanes)):
            step(lane_i)
        join_streams(dev)
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step(_)
    join_streams(dev)
    if dist is not None:
        # warm the collective the timed region ends with (RCCL sets up its point-to-point channels on first
        # use): the same gather on a small state, part of the warm-up like the W untimed steps
        from quantool_amd.engine.sharding import gather_state_dict

        gather_state_dict({f"warm.{rank}": torch.zeros(1 << 20, dtype=torch.uint8, device=dev)}, dst=0,
                          device="cpu" if REHEARSE else dev)
    barrier()

    lib.qt_profile_enable(1)
    kept = []
    barrier()
    mem0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kept.append(step(_))
    t_host = time.perf_counter() - t0             # host time to enqueue the steps (the GPU runs behind it)
    join_streams(dev)
    torch.cuda.synchronize()
    t_compute = time.perf_counter() - t0          # this rank's own steps, before the collective
    mem1 = torch.cuda.memory_stats(dev)
    # hipMalloc / hipFree calls of the caching allocator INSIDE the timed region (each one is a host stall, a hipFree a
    # device synchronisation): 0 when the warm-up steps have sized every pool
    allocator = {k: int(mem1.get(k, 0) - mem0.get(k, 0)) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries")}
    if dist is not None:
        # final gather of the packed state to rank 0 (the job's only collective)
        from quantool_amd.engine.sharding import gather_state_dict

        local = {f"layers.{rank + world * i}.{k}": v for i, outs in enumerate(kept) for k, v in outs.items()}
        merged = gather_state_dict(local, dst=0, device="cpu" if REHEARSE else dev)
        if rank == 0:
            assert len(merged) == len(local) * world
    barrier()
    elapsed = time.perf_counter() - t0

    tot_ms = ctypes.c_double()
    launches = ctypes.c_int64()
    lib.qt_profile_read(_lib_const("QT_PROF_XTX"), ctypes.byref(tot_ms), ctypes.byref(launches))
    lib.qt_profile_enable(0)

    # the same kernel with nothing else in flight (outside the timed region): under the overlapped
    # schedule the live launch duration includes time spent sharing CUs with the other streams' kernels
    iso_ms, iso_flops = 0.0, 0.0
    if rank == 0:
        from quantool_amd.hip import ops as _ops

        lib.qt_profile_enable(1)
        by_k = {}
        for gname, K, _ in shape.groups:      # one group per distinct in_features: the first (full token count)
            by_k.setdefault(K, gname)
        for K, gname in by_k.items():
            Gtmp = torch.zeros((K, K), dtype=torch.float32, device=dev)
            for _ in range(2):
                _ops.xtx_accumulate(acts[gname], Gtmp)
                iso_flops += acts[gname].shape[0] * K * (K + 1)
            torch.cuda.synchronize()
            del Gtmp
        t_iso, n_iso = ctypes.c_double(), ctypes.c_int64()
        lib.qt_profile_read(_lib_const("QT_PROF_XTX"), ctypes.byref(t_iso), ctypes.byref(n_iso))
        lib.qt_profile_enable(0)
        iso_ms = t_iso.value

    t_max = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if REHEARSE else dev)
    per_rank = [t_compute]
    if dist is not None:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        mine = torch.tensor([t_compute], dtype=torch.float64, device=t_max.device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
    elapsed = float(t_max.item())

    weights_done = shape.weights_per_layer * args.steps * world
    value = weights_done / elapsed

    # roofline of the dominant kernel (xtx_kernel): algorithmic flops N*K*(K+1) per Hessian
    # (symmetric minimum) / device time of the kernel from HIP events on its stream
    alg_flops = sum(acts[g].shape[0] * K * (K + 1) for g, K, _ in shape.groups) * args.steps
    alg_bytes = sum(acts[g].shape[0] * K * 2 + K * K * 4 for g, K, _ in shape.groups) * args.steps
    if awq:   # + one D^T D Gram pass per grid point and balance Linear (rows play the tokens' part)
        alg_flops += sum(20 * R * K * (K + 1) for _, K, lins in shape.groups for _, R in lins) * args.steps
        alg_bytes += sum(20 * (R * K * 2 + K * K * 4) for _, K, lins in shape.groups for _, R in lins) * args.steps
    achieved_tflops = alg_flops / (tot_ms.value * 1e-3) / 1e12 if tot_ms.value > 0 else 0.0
    roofline = {
        "kernel": "xtx_kernel / xtx16_kernel (the Gram kernel on its two MFMA shapes)", "bound": "mfma", "achieved": round(achieved_tflops, 2),
        "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved_tflops / PEAK_BF16_MFMA_TFLOPS, 4),
        "traffic": pmc_traffic() if (args.model == "llama-3-8b" and not awq) else None,
        "launches": int(launches.value), "avg_launch_ms": round(tot_ms.value / max(1, launches.value), 4),
        "hbm_GBps_algorithmic": round(alg_bytes / (tot_ms.value * 1e-3) / 1e9, 1) if tot_ms.value > 0 else 0.0,
        "achieved_isolated": round(iso_flops / (iso_ms * 1e-3) / 1e12, 2) if iso_ms > 0 else None,
        "frac_isolated": round(iso_flops / (iso_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4) if iso_ms > 0 else None,
        "note": ("achieved/frac: live in the timed region, flop-weighted over the step's Gram launches (one per "
                 "distinct input: " + ", ".join(f"{sum(1 for _, k, _ in shape.groups if k == K)}x K={K}"
                                                for K in sorted({k for _, k, _ in shape.groups})) + "); the groups of "
                 f"a layer and {args.lanes} layers run concurrently on separate streams, so a live launch shares the CUs with "
                 "other kernels. *_isolated: one launch per distinct K (2 repetitions) alone on the GPU right after "
                 "the timed region. traffic: from the committed PMC profile of the Llama-3-8B workload (bench.py "
                 "cannot run the profiler). north_star's HBM figure is hbm_GBps_algorithmic; X^T X is MFMA-bound "
                 "(SURVEY 8d)"),
    }

    stages = stage_rates = None
    if rank == 0 and world == 1 and not awq and not args.no_stage_split:
        stages, stage_rates = stage_split(shape, weights, acts, args.samples, smooth)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not awq and args.model == "llama-3-8b":
        cpu = cpu_baseline_port()

    if rank == 0:
        wpl = shape.weights_per_layer
        n_lin = sum(len(l) for _, _, l in shape.groups)
        if awq:
            metric = ("quantized weights/sec (AWQ int4, Llama-3-8B, 512 calib samples) "
                      "[BASELINE.json configs[2], not the headline]")
            workload = ("Llama-3-8B-shaped random-init AWQ int4 g128 (W4A16, 20-point per-channel scale search, "
                        "duo scaling), 512 synthetic calib samples x 384 tokens, 1 decoder layer (4 mappings, "
                        f"7 Linears, {wpl} weights) per step per GPU")
        elif args.model == "llama-3-70b":
            metric = ("quantized weights/sec (GPTQ int4, Llama-3-70B, 512 calib samples) "
                      "[BASELINE.json configs[3] on one GPU, not the headline]")
            workload = (f"Llama-3-70B-shaped random-init GPTQ int4 g128 (W4A16, actorder={args.actorder}, dampening 0.01, "
                        "block 128), 512 synthetic calib samples x 384 tokens, 1 decoder layer (hidden 8192, inter 28672; "
                        f"{n_lin} Linears, {wpl} weights) per step per GPU")
        elif args.model == "mixtral-8x7b":
            metric = ("quantized weights/sec (SmoothQuant + GPTQ int4, Mixtral-8x7B, 512 calib samples) "
                      "[BASELINE.json configs[4] on one GPU, not the headline]")
            workload = (f"Mixtral-8x7B-shaped random-init SmoothQuant (q/k/v <- input norm, strength 0.5) + GPTQ int4 g128 "
                        f"(W4A8 preset weights, actorder={args.actorder}) on every Linear incl. the 8 experts, 512 synthetic "
                        "calib samples x 384 tokens routed top-2 by a seeded router (tokens per expert: "
                        f"{min(route_counts)}..{max(route_counts)}), 1 decoder layer ({n_lin} Linears, {wpl} weights) "
                        "per step per GPU")
        else:
            metric = "quantized weights/sec (GPTQ int4, Llama-3-8B, 512 calib samples)"
            workload = (f"Llama-3-8B-shaped random-init GPTQ int4 g128 (W4A16, actorder={args.actorder}, "
                        "dampening 0.01, block 128), 512 synthetic calib samples x 384 tokens, "
                        f"1 decoder layer (7 Linears, {wpl} weights) per step per GPU")
        if SKIP:
            metric = f"DIAGNOSTIC (QT_BENCH_SKIP={SKIP}: half of the step left out) -- not a measurement of: " + metric
        if SHAPE_SCALE > 1:
            metric = f"REHEARSAL (QT_BENCH_SHAPE_SCALE={SHAPE_SCALE}: layer widths divided) -- not a measurement of: " + metric
        line = {
            "metric": metric,
            "value": value, "unit": "weights/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "per_rank_compute_ms_per_step": [round(t / args.steps * 1e3, 3) for t in per_rank],
            "gather_ms": round((elapsed - max(per_rank)) * 1e3, 3) if dist is not None else 0.0,
            "host_enqueue_ms_per_step": round(t_host / args.steps * 1e3, 3),
            "layers_in_flight": args.lanes,
            "allocator_calls_in_timed_region": allocator,
            "peak_hbm_GiB": {"allocated": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                             "reserved": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 1)},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" + (" (gloo rehearsal, ranks share one GPU)" if REHEARSE else ""),
            "config": {
                "workload": workload, "model": args.model, "method": args.method,
                "n_calibration_samples": args.samples, "seq_len": SEQ_LEN, "accumulate": args.accumulate,
                "layers_per_step_per_gpu": 1, "sharding": f"layers over {world} rank(s), RCCL gather of packed state"
                                                       + (" (one-rank RCCL group)" if ONE_RANK_RCCL and world == 1 else ""),
            },
            "roofline": roofline,
            "stages_ms_isolated": stages,
            "stages_rate_isolated": stage_rates,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def pmc_traffic():
    """HBM bytes per xtx_kernel launch (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes
    over this same workload, committed under profiles/); bench.py cannot run the profiler itself."""
    for name in ("r04_xtx_pmc_traffic.json", "r03_xtx_pmc_traffic.json"):     # the newest collection of the unchanged kernel
        try:
            return float(json.loads((ROOT / "profiles" / name).read_text())["avg_bytes_per_launch_over_a_step"])
        except Exception:
            continue
    return None


def _lib_const(name: str) -> int:
    return {"QT_PROF_XTX": 0, "QT_PROF_SWEEP_BLOCK": 1}[name]


if __name__ == "__main__":
    main()
