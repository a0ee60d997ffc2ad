"""``oneshot``: the single call quantool's llm-compressor plugins make
(``self.last_model = oneshot_fn(**oneshot_kwargs)``, ``src/quantool/methods/llm_compressor/base.py:161``),
re-provided on top of the HIP per-Linear path.

Accepted ``model`` values
  * ``LinearCalibrationSet`` -- explicit (activations, weights) groups: the per-Linear mode the
    benchmarks and parity tests use (BASELINE.md 2.2 "per-linear microbenchmark");
  * a ``torch.nn.Module`` (or a local directory loadable with ``transformers``) -- the sequential
    decoder-layer pipeline (``engine.sequential``).

Returns an object with ``save_pretrained(dest, save_compressed=True)`` like the model upstream
returns (``base.py:188``).
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, Iterable, List, Optional, Tuple

import torch

from ..hip import ops
from .gptq_linear import (HessianAccumulator, batch_chains_enabled, batchable, gptq_quantize_batched,
                          gptq_quantize_shared)
from .modifiers import AWQModifier, GPTQModifier, SmoothQuantModifier

logger = logging.getLogger(__name__)


@dataclass
class LinearGroup:
    """Linears that read the same activation.  ``activations``: tensor [S, T, K] / [N, K] (bf16) or
    an iterable of such batches; ``weights``: name -> [R, K] tensor."""
    name: str
    activations: Any
    weights: Dict[str, torch.Tensor]
    num_samples: Optional[int] = None
    #: 1-d parameters of the op that PRODUCES ``activations`` (a norm's weight / bias), by name.  Given,
    #: the group is a SmoothQuant mapping: a recipe with a SmoothQuantModifier rescales weights, these
    #: vectors and the activations; absent, the group is left alone by the smoothing stage.
    smooth_vectors: Optional[Dict[str, torch.Tensor]] = None


@dataclass
class LinearCalibrationSet:
    groups: List[LinearGroup]
    model_name: str = "synthetic-linears"

    def parameters_quantized(self) -> int:
        return sum(w.numel() for g in self.groups for w in g.weights.values())


class QuantizedLinears:
    """Result of ``oneshot`` on a ``LinearCalibrationSet``: the compressed state_dict."""

    def __init__(self, results: Dict[str, Any], recipe, scheme_name: str, fmt: str, weight_config: dict,
                 ignore: List[str], smoothed: Optional[Dict[str, torch.Tensor]] = None,
                 smoothing_scales: Optional[Dict[str, torch.Tensor]] = None):
        self.results = results
        self.smoothed = smoothed or {}                    # norm weights / biases after the SmoothQuant stage
        self.smoothing_scales = smoothing_scales or {}    # group name -> s[K]
        self.recipe = recipe
        self.scheme_name = scheme_name
        self.format = fmt
        self.weight_config = weight_config
        self.ignore = ignore

    def state_dict(self) -> Dict[str, torch.Tensor]:
        from .serialization import result_tensors

        sd: Dict[str, torch.Tensor] = {}
        for name, r in self.results.items():
            for k, v in result_tensors(r).items():
                sd[f"{name}.{k}"] = v
        sd.update(self.smoothed)
        return sd

    def quantization_config(self) -> dict:
        from .serialization import quantization_config

        from .schemes import preset_name_to_scheme

        acts = preset_name_to_scheme(self.scheme_name).input_activations
        return quantization_config(self.weight_config, self.format, self.ignore,
                                   acts.to_config() if acts is not None else None)

    def save_pretrained(self, save_directory, save_compressed: bool = True, max_shard_size="5GB", **_):
        from .serialization import save_state

        save_state(self.state_dict(), self.quantization_config(), save_directory, max_shard_size=max_shard_size)


def _iter_batches(acts) -> Iterable[torch.Tensor]:
    if isinstance(acts, torch.Tensor):
        yield acts
    else:
        for a in acts:
            yield a


def _select_modifiers(recipe) -> Tuple[Optional[SmoothQuantModifier], Optional[GPTQModifier], Optional[AWQModifier]]:
    mods = recipe if isinstance(recipe, (list, tuple)) else [recipe]
    sq = next((m for m in mods if isinstance(m, SmoothQuantModifier)), None)
    gp = next((m for m in mods if isinstance(m, GPTQModifier)), None)
    aw = next((m for m in mods if isinstance(m, AWQModifier)), None)
    if gp is None and aw is None:
        raise ValueError("recipe must contain a GPTQModifier or an AWQModifier")
    return sq, gp, aw


def _oneshot_linears(cal: LinearCalibrationSet, recipe, device) -> QuantizedLinears:
    sq, gp, aw = _select_modifiers(recipe)
    results: Dict[str, Any] = {}
    smoothed: Dict[str, torch.Tensor] = {}
    smoothing_scales: Dict[str, torch.Tensor] = {}
    if gp is not None:
        qargs = gp.weight_args()
        from .sharding import (GatheredResult, allreduce_accumulator, dist_world, gather_state_dict, group_cost,
                               gptq_quantize_row_split, plan_groups)
        from .streams import GroupStreams

        # One process per GPU (torchrun): every rank is given the same calibration set; whole groups go to
        # ranks LPT-greedy (partitioning A), a group heavier than a fair share is split over all ranks
        # (partitioning B: tokens for the Gram sum, rows for the sweep), and rank 0 ends up with the
        # whole quantised state (north_star: "RCCL ... only to gather the final state_dict").
        world, rank = dist_world()
        order = sorted(cal.groups, key=lambda g: (-int(next(iter(g.weights.values())).shape[1]), g.name))
        if world > 1:
            def n_tok(g):
                a = g.activations
                if isinstance(a, torch.Tensor):
                    return a.numel() // a.shape[-1]
                if isinstance(a, (list, tuple)):
                    return sum(t.numel() // t.shape[-1] for t in a)
                return 196608     # an iterator cannot be sized without consuming it: the default calibration set
            costs = [group_cost(int(next(iter(g.weights.values())).shape[1]), n_tok(g),
                                sum(int(w.shape[0]) for w in g.weights.values())) for g in order]
            plan = dict(zip((g.name for g in order), plan_groups(costs, world)))
        else:
            plan = {g.name: ("A", 0) for g in order}

        def local_batches(g, split: bool):
            """This rank's calibration batches of a group, with the sample count they stand for."""
            if not split:
                for xb in _iter_batches(g.activations):
                    yield xb, None
                return
            a = g.activations
            if isinstance(a, torch.Tensor) and a.dim() == 3:
                for i in range(rank, a.shape[0], world):
                    yield a[i:i + 1], 1
            elif isinstance(a, torch.Tensor):
                rows = a.reshape(-1, a.shape[-1])
                per = (rows.shape[0] + world - 1) // world
                yield rows[rank * per:(rank + 1) * per], (1 if rank == 0 else 0)
            else:
                for i, xb in enumerate(_iter_batches(a)):
                    if i % world == rank:
                        yield xb, None

        def gram_phase(g):
            """Smoothing stage and Gram sum of one group.  Returns what `chain_phase` needs, or None when this
            rank has nothing to do for the group."""
            kind, owner = plan[g.name]
            if kind == "A" and owner != rank:
                return None
            weights = {n: w.to(device) for n, w in g.weights.items()}
            rescale = None
            if sq is not None and g.smooth_vectors:
                # SmoothQuant stage of the recipe: every Linear of the group is a balance
                # layer -- ignored ones included, they read the same rescaled activation.  (A split group
                # computes the same scales from the full activations on every rank.)
                rescale = _smooth_group(g, weights, sq.smoothing_strength, device, smoothed)
                smoothing_scales[g.name] = rescale
            names = [n for n in weights if n.split(".")[-1] not in gp.ignore and n not in gp.ignore]
            if not names:
                return None
            K = weights[names[0]].shape[1]
            acc = HessianAccumulator(K, device)
            for xb, ns in local_batches(g, kind == "B"):
                xb = xb.to(device)
                if xb.numel() == 0:
                    continue
                if rescale is not None:      # what the smoothed norm now emits: X / s, in the activation dtype
                    xb = ops.scale_columns(xb.reshape(-1, K), rescale, divide=True).reshape(xb.shape)
                acc.add(xb, num_samples=ns)
            acc.flush()                      # staged tokens go through the Gram kernel here, not on the chain's stream
            if kind == "B":
                allreduce_accumulator(acc)
            if g.num_samples is not None:
                acc.n = int(g.num_samples)
            return kind, names, [weights[n] for n in names], acc

        def chain_phase(state):
            kind, names, ws, acc = state
            if kind == "B":
                res = gptq_quantize_row_split(ws, acc, qargs, block_size=gp.block_size, dampening_frac=gp.dampening_frac)
            else:
                res = gptq_quantize_shared(ws, acc, qargs, block_size=gp.block_size, dampening_frac=gp.dampening_frac)
            results.update(dict(zip(names, res)))

        if world > 1:
            for g in order:          # collectives must be issued in the same order on every rank: one stream
                state = gram_phase(g)
                if state is not None:
                    chain_phase(state)
            from .serialization import result_tensors

            mine = {f"{n}::{k}": v for n, r in results.items() if plan_owner_is(plan, cal, n, rank)
                    for k, v in result_tensors(r).items()}
            # the SmoothQuant stage's outputs travel with the owner's results: the rescaled producer vectors
            # (norm weight / bias divided by s) and the scales exist only where `_smooth_group` ran, and rank 0
            # must not save W*s for a group without its v/s
            for g in cal.groups:
                kind, owner = plan[g.name]
                if (owner if kind == "A" else 0) != rank or g.name not in smoothing_scales:
                    continue
                mine[f"{g.name}::@smoothing_scale"] = smoothing_scales[g.name]
                for vn in g.smooth_vectors:
                    mine[f"{vn}::@smoothed"] = smoothed[vn]
            if torch.distributed.get_backend() == "gloo":     # host-side backend (CPU rehearsals): ship host tensors
                mine = {k: v.cpu() for k, v in mine.items()}
                merged = gather_state_dict(mine, dst=0, device=None)
            else:
                merged = gather_state_dict(mine, dst=0, device=device)
            if rank == 0:
                by_lin: Dict[str, dict] = {}
                for key, t in merged.items():
                    n, k = key.split("::")
                    if k == "@smoothing_scale":
                        smoothing_scales.setdefault(n, t.to(device))
                    elif k == "@smoothed":
                        smoothed.setdefault(n, t.to(device))
                    else:
                        by_lin.setdefault(n, {})[k] = t.to(device)
                for n, parts in by_lin.items():
                    if n not in results:
                        if "weight" in parts:
                            parts["weight_q"] = parts.pop("weight")
                        results[n] = GatheredResult(parts, None)
        else:
            # the Gram sums on one stream, smallest in_features first; each group's chain on a stream of its own
            # behind its Gram sum: see streams.py
            # ... when its activations arrive in launches of their own (>= DIRECT_TOKENS rows).  Small batches are
            # staged by many tiny copies, which one stream would put end to end for all groups (measured: 512 x
            # 384-token calls per group, 86.3 vs 83.3 ms per Llama-3-8B layer): those groups keep their Gram sum on
            # their own stream.
            def long_batches(g) -> bool:
                a = g.activations
                if isinstance(a, torch.Tensor):
                    a = [a]
                if not isinstance(a, (list, tuple)) or not a:
                    return False
                return all(isinstance(t, torch.Tensor) and t.numel() // max(1, t.shape[-1]) >= HessianAccumulator.DIRECT_TOKENS
                           for t in a)

            def chain_phase_batched(states):
                """Groups of equal in_features through one chain of launches (gptq_quantize_batched; bit-identical per
                group to chain_phase)."""
                res = gptq_quantize_batched([(ws, acc) for _, _, ws, acc in states], qargs, block_size=gp.block_size,
                                            dampening_frac=gp.dampening_frac)
                for (_, names, _, _), rs in zip(states, res):
                    results.update(dict(zip(names, rs)))

            pool = GroupStreams(device)
            behind_gram = []      # (state, event): chains whose Gram sum runs on the Gram stream
            for g in sorted(order, key=lambda g: int(next(iter(g.weights.values())).shape[1])):
                if not long_batches(g):
                    def both(g=g):
                        state = gram_phase(g)
                        if state is not None:
                            chain_phase(state)
                    pool.run(both)
                    continue
                state, ready = pool.run_gram(lambda g=g: gram_phase(g))
                if state is None:
                    continue
                if batch_chains_enabled():
                    behind_gram.append((state, ready))
                else:                      # issued right behind its Gram sum: the host never runs a layer ahead
                    pool.run(lambda state=state: chain_phase(state), after=ready,
                             tensors=[state[3].G] + list(state[2]))
            # the groups of equal in_features share one batched factorisation and one stacked sweep, on one stream
            # behind the Gram sums of all of them
            for idx in batchable([(st[2], st[3]) for st, _ in behind_gram]):
                members = [behind_gram[i] for i in idx]
                states = [st for st, _ in members]
                tensors = [t for st in states for t in [st[3].G] + list(st[2])]
                if len(states) == 1:
                    pool.run(lambda state=states[0]: chain_phase(state), after=members[0][1], tensors=tensors)
                else:
                    pool.run(lambda states=states: chain_phase_batched(states), after=[ev for _, ev in members], tensors=tensors)
            pool.join()
        mod = gp
    else:
        from .awq_linear import awq_quantize_groups

        qargs = aw.weight_args()
        todo = []
        for g in cal.groups:
            names = [n for n in g.weights if n.split(".")[-1] not in aw.ignore and n not in aw.ignore]
            if names:
                todo.append((names, [g.weights[n].to(device) for n in names], _iter_batches(g.activations)))
        # every mapping's search is enqueued before the host waits for the first (awq_quantize_groups); in chunks of
        # eight mappings, so that at most eight Gram matrices are alive at once
        for c in range(0, len(todo), 8):
            chunk = todo[c:c + 8]
            for (names, _, _), res in zip(chunk, awq_quantize_groups([(w, b) for _, w, b in chunk], qargs, n_grid=aw.n_grid,
                                                                     duo_scaling=aw.duo_scaling, device=device)):
                results.update(dict(zip(names, res)))
        mod = aw
    return QuantizedLinears(results, recipe, mod.scheme, mod.resolved_scheme.format, qargs.to_config(), list(mod.ignore),
                            smoothed, smoothing_scales)


def plan_owner_is(plan, cal, lin_name: str, rank: int) -> bool:
    """True when ``rank`` is the one that sends Linear ``lin_name`` in the final gather: the owner of its
    group under partitioning A, rank 0 for a split group (every rank holds those)."""
    for g in cal.groups:
        if lin_name in g.weights:
            kind, owner = plan[g.name]
            return owner == rank if kind == "A" else rank == 0
    return False


def _smooth_group(g: LinearGroup, weights: Dict[str, torch.Tensor], alpha: float, device,
                  smoothed: Dict[str, torch.Tensor]) -> torch.Tensor:
    """s = (max - min of the activation)^alpha / (max |W| over the group's Linears)^(1-alpha);
    ``weights`` are replaced by W * s in place of the dict, the producing vectors by v / s."""
    from .smoothquant import ChannelMinMax, apply_smoothing, smoothquant_scales

    names = list(weights)
    K = weights[names[0]].shape[1]
    stats = ChannelMinMax(K, device)
    for xb in _iter_batches(g.activations):
        stats.add(xb.to(device).reshape(-1, K))
    s = smoothquant_scales(stats, [weights[n] for n in names], alpha)
    vec_names = list(g.smooth_vectors)
    new_w, new_v = apply_smoothing(s, [weights[n] for n in names], [g.smooth_vectors[n].to(device) for n in vec_names])
    weights.update(zip(names, new_w))
    smoothed.update(zip(vec_names, new_v))
    return s


def oneshot(model=None, dataset=None, recipe=None, output_dir: Optional[str] = None,
            num_calibration_samples: int = 512, max_seq_length: int = 384,
            shuffle_calibration_samples: bool = True, save_compressed: bool = True,
            trust_remote_code_model: bool = False, dataset_path: Optional[str] = None,
            calibration_dataloader=None, tokenizer=None, processor=None, splits=None,
            dataset_config_name: Optional[str] = None, text_column: str = "text", pad_to_max_length: bool = False,
            # upstream names that carry meaning here
            precision="auto", sequential_targets=None, pipeline: Optional[str] = None,
            # upstream names accepted so that quantool routes them here instead of dropping them
            # (base.py:117-124 matches kwargs against this signature); they concern hub access, logging or
            # sparsity stages or MoE calibration policy (experts are always calibrated on their routed tokens only),
            # none of which exists in this backend
            min_tokens_per_module: Optional[float] = None, calibrate_moe_context: bool = False,
            config_name=None, cache_dir=None, use_auth_token=False, tie_word_embeddings=False,
            model_revision: str = "main", recipe_args=None, clear_sparse_session: bool = False, stage=None,
            concatenate_data: bool = False, streaming: bool = False, overwrite_cache: bool = False,
            preprocessing_num_workers=None, tracing_ignore=None, quantization_aware_calibration: bool = True,
            log_dir: Optional[str] = None,
            device: Optional[str] = None, seed: int = 42, **unused):
    """Counterpart of ``llmcompressor.oneshot`` for the GPTQ / AWQ / SmoothQuant recipes.

    Keyword names are upstream's ([UPSTREAM-RECALL]), because quantool routes
    ``quantize(**kwargs)`` entries here by matching them against this signature
    (``base.py:45-72,117-124``).  ``precision`` selects the dtype a model PATH is loaded in ("auto" =
    the checkpoint's own); ``sequential_targets`` names the blocks calibrated one after another;
    ``pipeline`` must be sequential-compatible ("sequential", "independent", "basic" or None: this
    backend always calibrates block by block).
    """
    if pipeline not in (None, "sequential", "independent", "basic", "datafree"):
        raise ValueError(f"pipeline={pipeline!r}: expected 'sequential', 'independent', 'basic' or None")
    ignored = {k: v for k, v in dict(config_name=config_name, cache_dir=cache_dir, recipe_args=recipe_args, stage=stage,
                                     log_dir=log_dir, tracing_ignore=tracing_ignore,
                                     min_tokens_per_module=min_tokens_per_module,
                                     calibrate_moe_context=calibrate_moe_context).items() if v not in (None, False)}
    if ignored or unused:
        logger.info(f"oneshot: arguments without effect in this backend: {sorted(ignored) + sorted(unused)}")
    if recipe is None:
        raise ValueError("oneshot requires a recipe")
    from ..hip import _lib

    _lib.load()  # no CPU fallback: fail before touching the model if the HIP library is missing
    if not torch.cuda.is_available():
        raise RuntimeError("quantool_amd.oneshot needs an AMD GPU (torch.cuda is not available); "
                           "there is no CPU path")
    dev = torch.device(device or f"cuda:{torch.cuda.current_device()}")

    if isinstance(model, LinearCalibrationSet):
        out = _oneshot_linears(model, recipe, dev)
    else:
        from .sequential import oneshot_module

        out = oneshot_module(model, dataset, recipe, dev, num_calibration_samples=num_calibration_samples,
                             max_seq_length=max_seq_length, shuffle=shuffle_calibration_samples, tokenizer=tokenizer,
                             dataloader=calibration_dataloader, dataset_path=dataset_path, text_column=text_column,
                             trust_remote_code=trust_remote_code_model, seed=seed, precision=precision,
                             sequential_targets=sequential_targets)
    # scratch buffers are cached per (device, stream, purpose) while a job runs: give them back now
    ops.release_workspaces()
    if output_dir:
        from .sharding import dist_world

        if dist_world()[1] == 0:      # under torchrun rank 0 holds the whole state and is the one that writes
            Path(output_dir).mkdir(parents=True, exist_ok=True)
            out.save_pretrained(str(output_dir), save_compressed=save_compressed)
    return out
