"""Sequential calibration driver for ``torch.nn.Module`` models: what
``llmcompressor.oneshot`` does under ``base.py:161`` for a decoder-only transformer, on top of the
HIP per-Linear path.

Per decoder layer: (i) run every calibration batch through the layer with hooks on the targeted
Linears, accumulating one Gram matrix per *distinct input tensor* (q/k/v and gate/up share);
(ii) quantise the layer's Linears and write the dequantised weights back; (iii) re-run the layer
with the quantised weights to produce the next layer's inputs.
"""
from __future__ import annotations

import logging
import os
import random
import types
from pathlib import Path
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from .gptq_linear import (HessianAccumulator, batch_chains_enabled, batchable, gptq_quantize_batched,
                          gptq_quantize_shared)
from .modifiers import AWQModifier, GPTQModifier, SmoothQuantModifier
from .streams import GroupStreams

logger = logging.getLogger(__name__)


class _StopForward(Exception):
    pass


#: Test hook: when set to a dict, every input group's stage boundaries (factor U, permutation, dead
#: mask -- ``gptq_quantize_shared(keep=...)``) are stored under its leader's module name, so a parity
#: test can hand the very same factor to the oracle.  None in production (U is K x K fp32 per group).
DEBUG_KEEP: Optional[dict] = None


def find_decoder_layers(model: nn.Module, sequential_targets=None) -> List[nn.Module]:
    """The blocks calibrated one after another.

    ``sequential_targets`` (forwarded by the reference plugin, ``gptq/gptq.py:82-84``; upstream: class
    names such as ``"LlamaDecoderLayer"`` or module names) wins when given.  Otherwise the OUTERMOST
    homogeneous ``nn.ModuleList`` is taken (``model.model.layers``, ``model.model.decoder.layers``):
    outermost, not longest -- a sparse-MoE layer holds a ModuleList of experts that can outnumber
    the decoder layers."""
    if sequential_targets:
        targets = [sequential_targets] if isinstance(sequential_targets, str) else list(sequential_targets)
        hits = [m for n, m in model.named_modules() if type(m).__name__ in targets or n in targets]
        # a target nested inside another hit is part of that hit
        inner = {id(c) for m in hits for c in m.modules() if c is not m}
        hits = [m for m in hits if id(m) not in inner]
        if not hits:
            raise ValueError(f"sequential_targets={targets} matches no module (class names or module names expected)")
        return hits
    lists = [(n, m) for n, m in model.named_modules()
             if isinstance(m, nn.ModuleList) and len(m) > 0 and len({type(x) for x in m}) == 1]
    outer = [(n, m) for n, m in lists if not any(n.startswith(pn + ".") for pn, _ in lists if pn != n)]
    if not outer:
        raise ValueError("could not locate the decoder layers (no homogeneous nn.ModuleList found); "
                         "pass sequential_targets")
    return list(max(outer, key=lambda nm: len(nm[1]))[1])


#: What the calibration hooks need to know about the forward in flight: how many samples it carries and how many
#: tokens each has.  Upstream feeds one sample per forward and counts one per forward that reaches a Linear
#: (``num_added``); with several samples per forward a flattened ``[tokens, K]`` input -- an expert's routed rows --
#: counts the samples those rows come from, which only the routing code knows (``_UnfusedExperts.forward``).
_CALIB_CTX: Dict[str, Any] = {"samples": 1, "tokens_per_sample": None}


class _UnfusedExperts(nn.Module):
    """Stand-in for a fused sparse-MoE expert bank (transformers >= 5: ``gate_up_proj [E, 2I, H]`` and
    ``down_proj [E, H, I]`` as 3-d parameters): every expert's two matrices become ``nn.Linear``
    modules whose weights are VIEWS of the fused parameters, so the calibration hooks see each
    expert's routed tokens and the quantised weights land in the original storage.  The forward is
    the fused module's own routing loop with the two ``F.linear`` calls replaced by the Linears."""

    def __init__(self, fused: nn.Module):
        super().__init__()
        gu, dn = fused.gate_up_proj, fused.down_proj
        self.num_experts = int(gu.shape[0])
        self.act_fn = fused.act_fn
        self.experts = nn.ModuleList()
        for e in range(self.num_experts):
            blk = nn.Module()
            blk.gate_up_proj = nn.Linear(gu.shape[2], gu.shape[1], bias=False, device="meta")
            blk.down_proj = nn.Linear(dn.shape[2], dn.shape[1], bias=False, device="meta")
            blk.gate_up_proj.weight = nn.Parameter(gu.data[e], requires_grad=False)
            blk.down_proj.weight = nn.Parameter(dn.data[e], requires_grad=False)
            self.experts.append(blk)

    def forward(self, hidden_states, top_k_index, top_k_weights):
        out = torch.zeros_like(hidden_states)
        mask = torch.nn.functional.one_hot(top_k_index, num_classes=self.num_experts).permute(2, 1, 0)
        ctx = _CALIB_CTX
        batch, per = ctx["samples"], ctx["tokens_per_sample"]
        routed = mask.sum(dim=1) > 0                                  # [experts, tokens]
        n_tok = int(routed.shape[1])
        if batch > 1 and per:
            # per expert, the samples (of this forward) that routed at least one token to it -- what one forward per
            # sample would count -- for ALL experts in one device op and one host read (it was a torch.unique and a
            # host synchronisation per expert per forward)
            sid = torch.div(torch.arange(n_tok, device=routed.device), per, rounding_mode="floor")
            per_sample = torch.zeros((self.num_experts, (n_tok + per - 1) // per), dtype=torch.float32, device=routed.device)
            per_sample.index_add_(1, sid, routed.to(torch.float32))
            counts = (per_sample > 0).sum(dim=1).tolist()
        else:
            counts = [batch if hit else 0 for hit in routed.any(dim=1).tolist()]
        try:
            for e, n_samples in enumerate(counts):
                if n_samples == 0:
                    continue
                pos, tok = torch.where(mask[e])
                ctx["samples"] = int(n_samples)
                gate, up = self.experts[e].gate_up_proj(hidden_states[tok]).chunk(2, dim=-1)
                y = self.experts[e].down_proj(self.act_fn(gate) * up) * top_k_weights[tok, pos, None]
                out.index_add_(0, tok, y.to(out.dtype))
        finally:
            ctx["samples"] = batch       # also when an expert raised: the next layer / model starts from the forward's count
        return out


#: Per-expert checkpoint vocabulary by ``config.model_type``: (name of the MoE block in the per-expert layout or None to
#: keep the module's own, leaf names of the gate / up / down Linears).  Mixtral's per-expert checkpoints -- the layout
#: transformers' own loader converts FROM (``transformers/conversion_mapping.py``, entry "mixtral":
#: ``.block_sparse_moe.`` -> ``.mlp.``, ``.experts.*.w1.weight`` + ``.experts.*.w3.weight`` -> ``.experts.gate_up_proj``,
#: ``.experts.*.w2.weight`` -> ``.experts.down_proj``) -- say ``block_sparse_moe.experts.{e}.w1 / w3 / w2``; the
#: Qwen-MoE family (and the default) says ``mlp.experts.{e}.gate_proj / up_proj / down_proj``.
EXPERT_LAYOUTS = {
    "mixtral": ("block_sparse_moe", {"gate_proj": "w1", "up_proj": "w3", "down_proj": "w2"}),
    None: (None, {"gate_proj": "gate_proj", "up_proj": "up_proj", "down_proj": "down_proj"}),
}


def expert_layout(model_type):
    return EXPERT_LAYOUTS.get(model_type, EXPERT_LAYOUTS[None])


def expert_bank_module_renames(banks, model_type=None) -> Dict[str, str]:
    """{module-name prefix as the model has it: prefix as the checkpoint says it} for the MoE blocks that hold an
    unfused bank -- applied to tensor names AND to the module names inside ``quantization_config`` (``ignore``)."""
    block, _ = expert_layout(model_type)
    out = {}
    for bank in banks:
        parent, _, leaf = bank.rpartition(".")
        if block and parent and parent.rpartition(".")[2] != block:
            out[parent] = parent.rpartition(".")[0] + ("." if "." in parent else "") + block
    return out


def rename_module_prefix(name: str, renames: Dict[str, str]) -> str:
    for old, new in renames.items():
        if name == old or name.startswith(old + "."):
            return new + name[len(old):]
    return name


def expert_bank_checkpoint_names(state: Dict[str, torch.Tensor], banks: Dict[str, "_UnfusedExperts"],
                                 model_type=None) -> Dict[str, torch.Tensor]:
    """Checkpoint names for what ``_UnfusedExperts`` holds (it exists for calibration only; no loader knows
    its ``<bank>.experts.{e}.gate_up_proj`` modules).

    Written instead, per expert ``e`` of bank ``<bank>`` (e.g. ``model.layers.3.mlp.experts``) -- the per-expert
    Linear layout MoE checkpoints and their loaders use, which a compressed-tensors ``Linear`` target can address,
    in the vocabulary of the architecture (``EXPERT_LAYOUTS``: Mixtral ``block_sparse_moe.experts.{e}.w1 / w3 / w2``,
    otherwise ``experts.{e}.gate_proj / up_proj / down_proj``):

      ``<bank>.{e}.<gate>.*`` / ``<bank>.{e}.<up>.*``  rows ``[0, I)`` / ``[I, 2I)`` of the fused gate_up matrix:
                                 GPTQ rows are independent given the factor, so packed words, scales, zero-points
                                 split by rows exactly; ``weight_g_idx`` (per input column) is shared; ``weight_shape``
                                 becomes ``[I, H]``
      ``<bank>.{e}.<down>.*``     unchanged

    Dense (un-quantised or ``save_compressed=False``) expert weights are split the same way into
    ``....<gate>.weight`` / ``<up>.weight`` / ``<down>.weight``.  INTEGRATION.md section 5 documents it."""
    _, leafs = expert_layout(model_type)
    renames = expert_bank_module_renames(banks, model_type)
    out: Dict[str, torch.Tensor] = {}
    for name, t in state.items():
        bank = next((b for b in banks if name.startswith(b + ".experts.")), None)
        if bank is None:
            out[rename_module_prefix(name, renames)] = t
            continue
        e, proj, leaf = name[len(bank) + len(".experts."):].split(".", 2)
        cbank = rename_module_prefix(bank, renames)
        if proj != "gate_up_proj":
            out[f"{cbank}.{e}.{leafs.get(proj, proj)}.{leaf}"] = t
            continue
        gate, up = f"{cbank}.{e}.{leafs['gate_proj']}", f"{cbank}.{e}.{leafs['up_proj']}"
        if leaf == "weight_g_idx":                      # per input column: both halves read the same columns
            out[f"{gate}.{leaf}"] = t
            out[f"{up}.{leaf}"] = t.clone()          # safetensors refuses tensors that share storage
        elif leaf == "weight_shape":
            half = torch.tensor([int(t[0]) // 2, int(t[1])], dtype=t.dtype)
            out[f"{gate}.{leaf}"] = half
            out[f"{up}.{leaf}"] = half.clone()
        else:                                           # row-indexed: weight, weight_packed, weight_scale, weight_zero_point
            if t.shape[0] % 2:
                raise ValueError(f"{name}: {t.shape[0]} rows cannot be split into gate and up halves")
            inter = t.shape[0] // 2
            out[f"{gate}.{leaf}"] = t[:inter].contiguous()
            out[f"{up}.{leaf}"] = t[inter:].contiguous()
    return out


def unfuse_expert_banks(model: nn.Module) -> int:
    """Replace every fused expert bank by ``_UnfusedExperts``; returns how many were replaced."""
    n = 0
    for parent in list(model.modules()):
        for name, child in list(parent.named_children()):
            gu, dn = getattr(child, "gate_up_proj", None), getattr(child, "down_proj", None)
            if (isinstance(gu, nn.Parameter) and isinstance(dn, nn.Parameter) and gu.dim() == 3 and dn.dim() == 3
                    and hasattr(child, "act_fn") and not isinstance(child, _UnfusedExperts)):
                setattr(parent, name, _UnfusedExperts(child))
                n += 1
    return n


def uncovered_weight_fraction(layer: nn.Module, linears: Dict[str, nn.Module]) -> float:
    """Share of the layer's >= 2-d parameter elements that no targeted Linear holds."""
    covered = {id(m.weight) for m in linears.values()}
    storages = {m.weight.untyped_storage().data_ptr() for m in linears.values()}
    total = held = 0
    for p in layer.parameters():
        if p.dim() < 2:
            continue
        total += p.numel()
        if id(p) in covered or p.untyped_storage().data_ptr() in storages:
            held += p.numel()
    return 0.0 if total == 0 else 1.0 - held / total


def build_batches(dataset, tokenizer, num_samples: int, max_seq_length: int, shuffle: bool, seed: int,
                  text_column: str, dataloader=None) -> List[Dict[str, torch.Tensor]]:
    """Batch size 1, truncation to max_seq_length, no padding."""
    if dataloader is not None:
        out = []
        for b in dataloader:
            out.append(b if isinstance(b, dict) else {"input_ids": b})
            if len(out) >= num_samples:
                break
        return out
    if isinstance(dataset, (str, Path)):
        # upstream's oneshot(dataset="name") downloads the named dataset; this backend never fetches.
        # A local .json / .jsonl file of rows is read; anything else is refused instead of being
        # iterated character by character.
        path = Path(dataset)
        if path.is_file() and path.suffix in (".json", ".jsonl"):
            import json

            text = path.read_text(encoding="utf-8")
            dataset = json.loads(text) if path.suffix == ".json" else [json.loads(l) for l in text.splitlines() if l.strip()]
        else:
            raise ValueError(
                f"dataset={str(dataset)!r}: a dataset id cannot be resolved here (no hub access in this backend). "
                "Pass a datasets.Dataset, a list of rows, a local .json/.jsonl file, or calibration_dataloader.")
    if dataset is None:
        raise ValueError("no calibration data: pass dataset=, dataset_path= or calibration_dataloader=")
    if hasattr(dataset, "keys") and not hasattr(dataset, "column_names") and not isinstance(dataset, dict):
        dataset = dataset[list(dataset.keys())[0]]
    rows = list(dataset)
    idx = list(range(len(rows)))
    if shuffle:
        random.Random(seed).shuffle(idx)
    idx = idx[:num_samples]
    out = []
    for i in idx:
        row = rows[i]
        if isinstance(row, torch.Tensor):
            ids = row
        elif isinstance(row, dict) and "input_ids" in row:
            ids = torch.as_tensor(row["input_ids"])
        else:
            if isinstance(row, dict) and text_column in row:
                text = row[text_column]
            elif isinstance(row, str):
                text = row
            else:
                have = sorted(row) if isinstance(row, dict) else type(row).__name__
                raise ValueError(f"calibration row {i} is neither a tensor, a dict with 'input_ids' or "
                                 f"{text_column!r}, nor a string (got {have})")
            if tokenizer is None:
                raise ValueError("a tokenizer is required to calibrate on text rows")
            ids = torch.as_tensor(tokenizer(text, truncation=True, max_length=max_seq_length,
                                            add_special_tokens=True)["input_ids"])
        ids = ids.reshape(1, -1)[:, :max_seq_length].long()
        out.append({"input_ids": ids})
    return out


def stack_batches(batches: List[Dict[str, torch.Tensor]], max_tokens: int) -> List[Dict[str, torch.Tensor]]:
    """Consecutive calibration samples with the same keys and the same ``[1, T]``-shaped tensors, concatenated along
    the batch dimension up to ``max_tokens`` tokens per forward (``merge_cache`` does the same to cached layer inputs;
    the sample count is the batch dimension, so ``num_added`` stays one per sample)."""
    if max_tokens <= 0:
        return list(batches)

    def sig(b):
        return tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in b.items())) if all(
            torch.is_tensor(v) and v.dim() == 2 and v.shape[0] == 1 for v in b.values()) else None

    out, run = [], []

    def close():
        if len(run) == 1:
            out.append(run[0])
        elif run:
            out.append({k: torch.cat([r[k] for r in run], 0) for k in run[0]})
        run.clear()

    for b in batches:
        s_b = sig(b)
        tokens = next(iter(b.values())).shape[1] if s_b is not None else 0
        if run and (s_b is None or s_b != sig(run[0]) or (len(run) + 1) * tokens > max_tokens):
            close()
        if s_b is None:
            out.append(b)
        else:
            run.append(b)
    close()
    return out


def _to_dev(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(dev)
    if isinstance(x, (list, tuple)):
        return type(x)(_to_dev(v, dev) for v in x)
    if isinstance(x, dict):
        return {k: _to_dev(v, dev) for k, v in x.items()}
    return x


def _save_compressed(model: nn.Module, save_directory, save_compressed: bool = True, max_shard_size="5GB", **_):
    from .serialization import quantization_config, result_tensors, save_state

    from .sharding import dist_world

    if dist_world()[1] != 0:
        return            # under torchrun every rank holds the same quantised model; rank 0 writes it
    results: Dict[str, Any] = getattr(model, "_qt_results", {})
    meta = getattr(model, "_qt_meta", {})
    state: Dict[str, torch.Tensor] = {}
    quantized = set(results)
    banks = {n: m for n, m in model.named_modules() if isinstance(m, _UnfusedExperts)}
    for name, t in model.state_dict().items():
        mod = name.rsplit(".", 1)[0]
        if mod in quantized and name.endswith(".weight") and save_compressed:
            continue
        state[name] = t
    if save_compressed:
        for mod, r in results.items():
            for k, v in result_tensors(r).items():
                state[f"{mod}.{k}"] = v
    base_cfg = model.config.to_dict() if hasattr(model, "config") and hasattr(model.config, "to_dict") else {}
    ignore = list(meta.get("ignore", []))
    if banks:
        model_type = base_cfg.get("model_type")
        state = expert_bank_checkpoint_names(state, banks, model_type)
        # module names inside the saved quantization_config follow the tensors: same pass, same table
        renames = expert_bank_module_renames(banks, model_type)
        ignore = [rename_module_prefix(n, renames) for n in ignore]
    save_state(state, quantization_config(meta.get("weights", {}), meta.get("format", "pack-quantized"),
                                          ignore, meta.get("input_activations")), save_directory, base_cfg,
               max_shard_size=max_shard_size)


def oneshot_module(model, dataset, recipe, dev, *, num_calibration_samples: int, max_seq_length: int, shuffle: bool,
                   tokenizer=None, dataloader=None, dataset_path=None, text_column: str = "text",
                   trust_remote_code: bool = False, seed: int = 42, precision="auto", sequential_targets=None):
    mods = recipe if isinstance(recipe, (list, tuple)) else [recipe]
    gp = next((m for m in mods if isinstance(m, GPTQModifier)), None)
    sq = next((m for m in mods if isinstance(m, SmoothQuantModifier)), None)
    aw = next((m for m in mods if isinstance(m, AWQModifier)), None)
    if gp is None and aw is None:
        raise ValueError("recipe must contain a GPTQModifier or an AWQModifier")
    if gp is not None and aw is not None:
        raise ValueError("GPTQModifier and AWQModifier in one recipe: pick one weight quantizer")
    qm = gp if gp is not None else aw
    if isinstance(model, (str, Path)):
        from transformers import AutoModelForCausalLM, AutoTokenizer

        path = str(model)
        # the checkpoint's own dtype unless the caller says otherwise: the reference passes the model
        # path through with no dtype (base.py:222-241), upstream's `precision` defaults to "auto"
        table = {"auto": "auto", None: "auto", "float16": torch.float16, "fp16": torch.float16, "half": torch.float16,
                 "bfloat16": torch.bfloat16, "bf16": torch.bfloat16, "float32": torch.float32, "fp32": torch.float32,
                 "full": torch.float32}
        dt = precision if isinstance(precision, torch.dtype) else table.get(precision, precision)
        model = AutoModelForCausalLM.from_pretrained(path, dtype=dt, trust_remote_code=trust_remote_code,
                                                     local_files_only=True)
        if tokenizer is None:
            try:
                tokenizer = AutoTokenizer.from_pretrained(path, trust_remote_code=trust_remote_code,
                                                          local_files_only=True)
            except Exception:  # noqa: BLE001
                tokenizer = None
    if dataset is None and dataset_path is not None:
        import json

        with open(dataset_path, "r", encoding="utf-8") as fh:
            dataset = [json.loads(line) for line in fh if line.strip()]
    batches = build_batches(dataset, tokenizer, num_calibration_samples, max_seq_length, shuffle, seed, text_column,
                            dataloader)
    if not batches:
        raise ValueError("no calibration batches")
    # One process per GPU (torchrun): the calibration SAMPLES are split over the ranks (every rank holds
    # the model and forwards its own share), each input group's Gram sum is all-reduced once per layer,
    # every rank factorises the same Hessian and sweeps its slice of the rows, and the rows are
    # all-gathered -- partitioning B of SURVEY 8e for every group, so the replicated models stay equal.
    from .sharding import allreduce_accumulator, dist_world, gptq_quantize_row_split

    world, rank = dist_world()
    if world > 1 and aw is None:
        if len(batches) < world:
            raise ValueError(f"{len(batches)} calibration samples cannot be split over {world} ranks")
        batches = batches[rank::world]
    model.eval()
    model.to(dev)
    n_banks = unfuse_expert_banks(model)
    if n_banks:
        logger.info(f"unfused {n_banks} sparse-MoE expert bank(s) into per-expert Linears")
    seq_targets = sequential_targets or getattr(qm, "sequential_targets", None)
    layers = find_decoder_layers(model, seq_targets)
    qargs = qm.weight_args()

    # ---- inputs of the first decoder layer -------------------------------------------------------
    cache: List[tuple] = []

    def grab(_mod, args, kwargs):
        cache.append((args, kwargs))
        raise _StopForward

    ph = _Phases(dev)
    ph.start()
    h = layers[0].register_forward_pre_hook(grab, with_kwargs=True)
    batch_tokens = int(os.environ.get("QT_CALIB_BATCH_TOKENS", "32768"))
    with torch.no_grad():
        # equal-shape samples share a forward already here (the embedding is per token): 512 one-sample forwards up to
        # the first decoder layer were 0.2 s of a Llama-3-8B-sized job.  QT_CALIB_BATCH_TOKENS=0: one sample per forward.
        for b in stack_batches(batches, batch_tokens):
            try:
                model(**_to_dev(b, dev), use_cache=False)
            except _StopForward:
                pass
    h.remove()
    # whatever could not be stacked before the forward (extra inputs of unequal shape) is merged after it
    cache = merge_cache(cache, batch_tokens)
    ph.stop("first-layer inputs")

    prefix_of = {id(m): n for n, m in model.named_modules()}
    results: Dict[str, Any] = _ResultStore()
    with torch.no_grad():
        for li, layer in enumerate(layers):
            lname = prefix_of[id(layer)]
            if aw is not None:
                from .awq_module import awq_layer

                results.update(awq_layer(layer, lname, cache, aw, dev))
                if li + 1 < len(layers):
                    cache = _advance(layer, cache)
                continue
            linears = {f"{lname}.{n}" if n else lname: m for n, m in layer.named_modules()
                       if isinstance(m, nn.Linear) and gp.wants(f"{lname}.{n}", m)}
            miss = uncovered_weight_fraction(layer, linears)
            if miss > 0.10:
                logger.warning(f"{lname}: {100 * miss:.0f}% of the layer's matrix weights are held by no targeted "
                               "nn.Linear and stay dense (fused or custom-op weights?) although the saved config "
                               f"declares targets={list(qm.targets)}")
            if sq is not None:
                _smooth_layer(layer, cache, sq.smoothing_strength, dev, sq.mappings, lname)
            # Discovery on batch 0: which Linears read the same tensor.  This forward is ALSO batch 0's calibration
            # forward -- the inputs it shows are kept (that keeps them alive, too: a freed activation's address can be
            # handed to a later, unrelated tensor of the same shape, and pointer equality would then lie) and go into
            # the accumulators as soon as the grouping is known.  (Rounds 1-4 ran batch 0 twice: once to discover, once
            # to accumulate -- a sixth of a calibration pass per layer.)  It runs to its end: the call counts of a
            # whole forward decide whether the later ones may stop early.
            seen: Dict[str, List[torch.Tensor]] = {}      # every input a Linear was called with, in call order
            seen_version: Dict[str, List[int]] = {}       # ... and its version counter at that moment
            calls: Dict[str, int] = {}

            def discover(name):
                def fn(_m, a):
                    seen.setdefault(name, []).append(a[0])
                    seen_version.setdefault(name, []).append(a[0]._version)
                    calls[name] = calls.get(name, 0) + 1
                return fn

            ph.start()
            hooks = [m.register_forward_pre_hook(discover(n)) for n, m in linears.items()]
            args, kwargs = cache[0]
            layer(*args, **kwargs)
            for hk in hooks:
                hk.remove()
            groups: Dict[tuple, List[str]] = {}
            for n in linears:
                if n not in seen:
                    # not reached by batch 0 (a sparse-MoE expert none of its tokens was routed to):
                    # sharing cannot be established, so the Linear keeps a Hessian of its own
                    groups[("solo", n)] = [n]
                    continue
                t = seen[n][0]
                key = (t.untyped_storage().data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()))
                groups.setdefault(key, []).append(n)
            grouping = list(groups.values())
            if world > 1:
                # every rank discovered the sharing on ITS batch 0, and a Linear one rank's batch did not reach
                # (a sparse-MoE expert) would be a solo group there only: the ranks would then issue different
                # all-reduce sequences.  Rank 0's grouping is everybody's (a solo group is always valid: it
                # only forgoes the sharing).
                import torch.distributed as dist

                box = [grouping]
                dist.broadcast_object_list(box, src=0)
                grouping = box[0]
            leaders = {names[0]: names for names in grouping}
            accs = {lead: HessianAccumulator(linears[lead].in_features, dev) for lead in leaders}
            # upstream counts one sample per forward of a batch-size-1 pipeline (num_added); with several samples
            # per forward a [B, T, K] input counts B, and a flattened [tokens, K] input (OPT's fc1, routed expert
            # tokens) counts the samples of the forward it came from
            cur = _CALIB_CTX
            # A calibration forward is over once every input group has seen its rows: what the layer computes behind
            # its last hooked Linear (that Linear's own GEMM -- down_proj is a quarter of a Llama layer's flops -- and
            # the residual add) is recomputed by the propagate pass with the quantised weights anyway.  Only when the
            # discovery pass reached every targeted Linear exactly once (no weight shared between two call sites, no
            # expert left out by the router); a forward in which some group does not fire simply runs to its end.
            early_stop = (os.environ.get("QT_CALIB_EARLY_STOP", "1") != "0"
                          and all(calls.get(n, 0) == 1 for n in linears))
            fired = set()

            def add_hook(lead):
                def fn(_m, a):
                    x = a[0]
                    if x.dim() >= 3:
                        accs[lead].add(x.reshape(-1, x.shape[-2], x.shape[-1]))
                    else:
                        accs[lead].add(x.unsqueeze(0), num_samples=cur["samples"])
                    fired.add(lead)
                    if early_stop and len(fired) == len(leaders):
                        raise _StopForward
                return fn

            def set_batch(args):
                h0 = args[0] if args else None
                cur["samples"] = int(h0.shape[0]) if torch.is_tensor(h0) and h0.dim() >= 3 else 1
                cur["tokens_per_sample"] = int(h0.shape[1]) if torch.is_tensor(h0) and h0.dim() >= 3 else None

            # batch 0: what the discovery forward saw, group by group in the order the hooks would have fired
            # (QT_CALIB_MERGED_DISCOVERY=0: batch 0 is forwarded a second time, as up to round 4 -- A/B only)
            merged = os.environ.get("QT_CALIB_MERGED_DISCOVERY", "1") != "0"
            if merged and any(t._version != v for n in seen for t, v in zip(seen[n], seen_version[n])):
                # a kept input was written in place later in the forward (its version counter moved): what it holds now
                # is not what the Linear read.  Forward batch 0 again and take the inputs at hook time, as before.
                logger.info(f"{lname}: a Linear's input is modified in place behind it; batch 0 is forwarded twice")
                merged = False
            if merged:
                set_batch(cache[0][0])
                for lead in [n for n in seen if n in leaders]:
                    for x in seen[lead]:
                        if x.dim() >= 3:
                            accs[lead].add(x.reshape(-1, x.shape[-2], x.shape[-1]))
                        else:
                            accs[lead].add(x.unsqueeze(0), num_samples=cur["samples"])
            seen.clear()
            hooks = [linears[lead].register_forward_pre_hook(add_hook(lead)) for lead in leaders]
            for args, kwargs in (cache[1:] if merged else cache):
                set_batch(args)
                fired.clear()
                try:
                    layer(*args, **kwargs)
                except _StopForward:
                    pass
            for hk in hooks:
                hk.remove()
            for a_ in accs.values():
                a_.flush()
            ph.stop("calibration forwards + Gram")
            ph.start()
            # one stream per input group, largest in_features first (longest chain): see streams.py.  Also under
            # torchrun: the collectives inside a group's chain (the all-reduce of its Gram sum, the all-gather of its
            # rows) are issued by this one host thread in the same order on every rank whatever stream is current
            # (torch.distributed orders them on the process group's own stream behind an event of the current one);
            # QT_DIST_GROUP_STREAMS=0 puts everything back on one stream
            pool = GroupStreams(dev) if (world == 1 or os.environ.get("QT_DIST_GROUP_STREAMS", "1") != "0") else None

            def quantize_group(lead, names):
                ws = [linears[n].weight.data for n in names]
                if world > 1:
                    allreduce_accumulator(accs[lead])
                if accs[lead].n == 0:
                    # e.g. a sparse-MoE expert no calibration token was routed to.  Upstream would sweep
                    # with an all-zero Hessian (every column "dead", weights zeroed); keep the weights
                    # and round to nearest instead, loudly.
                    logger.warning(f"{names}: no calibration token reached this input; falling back to "
                                   "round-to-nearest for these Linears")
                    from .awq_linear import rtn_finalize

                    for n, w in zip(names, ws):
                        r = rtn_finalize(w, qargs)
                        linears[n].weight.data.copy_(r.dequantized(linears[n].weight.dtype))
                        results[n] = r
                    return
                keep = {} if DEBUG_KEEP is not None else None
                if world > 1:
                    res = gptq_quantize_row_split(ws, accs[lead], qargs, block_size=gp.block_size,
                                                  dampening_frac=gp.dampening_frac, with_dequantized=True)
                else:
                    res = gptq_quantize_shared(ws, accs[lead], qargs, block_size=gp.block_size,
                                               dampening_frac=gp.dampening_frac, keep=keep)
                if keep is not None and world == 1:
                    DEBUG_KEEP[lead] = dict(keep, names=list(names), n=accs[lead].n, G=accs[lead].G.clone())
                for n, r in zip(names, res):
                    linears[n].weight.data.copy_(r.dequantized(linears[n].weight.dtype))
                    results[n] = r

            def quantize_batch(leads):
                """Input groups of equal in_features through ONE chain of launches (gptq_quantize_batched): a Llama
                layer's q/k/v + o + gate/up, a sparse-MoE layer's experts.  Per group bit-identical to quantize_group."""
                keeps = [{} for _ in leads] if DEBUG_KEEP is not None else None
                res = gptq_quantize_batched([([linears[n].weight.data for n in leaders[lead]], accs[lead]) for lead in leads],
                                            qargs, block_size=gp.block_size, dampening_frac=gp.dampening_frac, keeps=keeps)
                for i, lead in enumerate(leads):
                    if keeps is not None:
                        DEBUG_KEEP[lead] = dict(keeps[i], names=list(leaders[lead]), n=accs[lead].n, G=accs[lead].G.clone())
                    for n, r in zip(leaders[lead], res[i]):
                        linears[n].weight.data.copy_(r.dequantized(linears[n].weight.dtype))
                        results[n] = r

            by_size = sorted(leaders.items(), key=lambda kv: (-linears[kv[0]].in_features, kv[0]))
            if world == 1 and pool is not None and batch_chains_enabled():     # (under ranks every group is row-split instead)
                live = [lead for lead, _ in by_size if accs[lead].n > 0]
                for lead, names in by_size:
                    if accs[lead].n == 0:
                        pool.run(lambda lead=lead, names=names: quantize_group(lead, names))      # the RTN fallback
                for idx in batchable([([linears[n].weight.data for n in leaders[lead]], accs[lead]) for lead in live]):
                    leads = [live[i] for i in idx]
                    if len(leads) == 1:
                        pool.run(lambda lead=leads[0]: quantize_group(lead, leaders[lead]))
                    else:
                        pool.run(lambda leads=leads: quantize_batch(leads))
            else:
                for lead, names in by_size:
                    if pool is not None:
                        pool.run(lambda lead=lead, names=names: quantize_group(lead, names))
                    else:
                        quantize_group(lead, names)
            # The chains are joined BEFORE the propagate pass.  (Tried in round 4: no join, every quantised Linear waiting
            # for its own chain's event in a forward pre-hook, so that a Llama layer's down_proj chain runs beside the
            # quantised layer's attention and gate / up GEMMs.  One of eight runs of test_gpu_config1_opt.py then
            # disagreed with the oracle in 6 packed words of the one Linear whose chain overlapped the pass; the cause
            # was not found, so the overlap is not shipped.)
            if pool is not None:
                pool.join()
            accs.clear()
            ph.stop("factorise + sweep + pack")
            ph.start()
            if li + 1 < len(layers):       # nobody reads the last layer's outputs: no propagate pass behind it
                cache = _advance(layer, cache)
            ph.stop("propagate")
            logger.info(f"quantized {lname}: {len(linears)} Linears in {len(leaders)} input groups")
    ph.report()
    model._qt_results = results
    acts = qm.resolved_scheme.input_activations
    model._qt_meta = {"weights": qargs.to_config(), "format": qm.resolved_scheme.format, "ignore": list(qm.ignore),
                      "input_activations": acts.to_config() if acts is not None else None}
    model.save_pretrained = types.MethodType(_save_compressed, model)
    return model


class _Phases:
    """QT_CALIB_TIMING=1: wall time per phase of the driver (device-synchronised; diagnostics only)."""

    def __init__(self, dev):
        self.on = os.environ.get("QT_CALIB_TIMING", "0") not in ("", "0")
        self.dev, self.t, self.acc = dev, 0.0, {}

    def start(self):
        if self.on:
            import time

            torch.cuda.synchronize(self.dev)
            self.t = time.perf_counter()

    def stop(self, name):
        if self.on:
            import time

            torch.cuda.synchronize(self.dev)
            self.acc[name] = self.acc.get(name, 0.0) + time.perf_counter() - self.t

    def report(self):
        if self.on:
            logger.warning("calibration phases: " + ", ".join(f"{k} {v:.3f} s" for k, v in self.acc.items()))


#: Beyond what the checkpoint needs (packed words, scales, zero points, g_idx, shape) a result object carries the
#: integer levels in sweep order and, for AWQ, the rescaled weight -- one to three bytes per weight, kept so that
#: ``dequantized()`` and the parity tests can look at them.  Over a whole model that is as large as the model itself
#: (Llama-3-70B-shaped: 68 GiB of levels next to 131 GiB of weights on one 288 GB GPU), so once the detail of the
#: results kept so far exceeds this many bytes, later results keep only their checkpoint tensors.
RESULT_DETAIL_BYTES = int(os.environ.get("QT_RESULT_DETAIL_BYTES", str(4 << 30)))


class _ResultStore(dict):
    """name -> result; strips the detail tensors of a result once the store holds RESULT_DETAIL_BYTES of them."""

    _DETAIL = ("Qt", "scaled_weight")

    def __init__(self):
        super().__init__()
        self.detail_bytes = 0

    def __setitem__(self, name, r):
        size = sum(t.numel() * t.element_size() for t in (getattr(r, f, None) for f in self._DETAIL)
                   if isinstance(t, torch.Tensor))
        if self.detail_bytes + size > RESULT_DETAIL_BYTES:
            for f in self._DETAIL:
                if isinstance(getattr(r, f, None), torch.Tensor):
                    setattr(r, f, None)      # dequantized() on such a result raises: the module holds that weight
        else:
            self.detail_bytes += size
        super().__setitem__(name, r)

    def update(self, other=(), **kw):
        for k, v in dict(other, **kw).items():
            self[k] = v


def merge_cache(cache: List[tuple], max_tokens: int) -> List[tuple]:
    """Stack consecutive cached layer inputs of identical structure along the batch dimension.

    The reference feeds the model one sample per forward; a decoder layer treats the rows of a batch
    independently (attention is per sample, MoE routing per token), so running B samples in one forward gives every
    hooked Linear the same rows -- up to the rounding of the layer's own GEMMs, whose tiling depends on the shape --
    with B-times fewer launches and far better GEMM shapes (512 x 384-token samples on one Llama-3-8B layer: 1.0 s ->
    see DESIGN 7).  Merged are tensors whose leading dimension is 1 (the batch dimension) and whose shapes agree;
    everything else (scalars, ``None``, tensors without a batch dimension such as ``cache_position``) must be equal
    across the merged samples and is taken from the first.  ``max_tokens <= 0`` keeps one sample per forward."""
    if max_tokens <= 0 or len(cache) < 2:
        return cache

    def leaves(x, path=()):
        if torch.is_tensor(x):
            yield path, x
        elif isinstance(x, (tuple, list)):
            for i, v in enumerate(x):
                yield from leaves(v, path + (i,))
        elif isinstance(x, dict):
            for k in sorted(x):
                yield from leaves(x[k], path + (k,))
        else:
            yield path, x

    def signature(entry):
        sig = []
        for path, v in leaves(entry):
            if torch.is_tensor(v):
                sig.append((path, "t", tuple(v.shape), v.dtype, v.dim() >= 2 and v.shape[0] == 1))
            else:
                sig.append((path, "o", repr(v)))
        return sig

    def compatible(a, b, sig_a):
        if sig_a != signature(b):
            return False
        for (pa, va), (_pb, vb) in zip(leaves(a), leaves(b)):
            if torch.is_tensor(va) and not (va.dim() >= 2 and va.shape[0] == 1) and not torch.equal(va, vb):
                return False     # a tensor without a batch dimension must be the same for every sample
        return True

    def stack(entries):
        def build(xs):
            x0 = xs[0]
            if torch.is_tensor(x0):
                return torch.cat(xs, 0) if (x0.dim() >= 2 and x0.shape[0] == 1) else x0
            if isinstance(x0, (tuple, list)):
                return type(x0)(build([x[i] for x in xs]) for i in range(len(x0)))
            if isinstance(x0, dict):
                return {k: build([x[k] for x in xs]) for k in x0}
            return x0
        return build(entries)

    def tokens(entry):
        h = entry[0][0] if entry[0] else None
        return int(h.shape[0] * h.shape[1]) if torch.is_tensor(h) and h.dim() >= 3 else 1 << 62

    out, run, run_sig, run_tok = [], [], None, 0
    for e in cache:
        t = tokens(e)
        if run and run_tok + t <= max_tokens and compatible(run[0], e, run_sig):
            run.append(e)
            run_tok += t
            continue
        if run:
            out.append(stack(run) if len(run) > 1 else run[0])
        run, run_sig, run_tok = [e], signature(e), t
    if run:
        out.append(stack(run) if len(run) > 1 else run[0])
    return out


def _advance(layer: nn.Module, cache):
    """Next layer's inputs: this layer re-run on its cached inputs with the quantised weights."""
    new_cache = []
    for args, kwargs in cache:
        out = layer(*args, **kwargs)
        out = out[0] if isinstance(out, (tuple, list)) else out
        new_cache.append(((out,) + tuple(args[1:]), kwargs))
    return new_cache


def _smooth_layer(layer: nn.Module, cache, alpha: float, dev, mappings=None, layer_name: str = "") -> None:
    """SmoothQuant pre-pass.  Default: each norm followed by the Linears that read its
    output ({q,k,v} <- input norm, {gate,up} / fc1 <- post-attention norm), discovered from tensor
    identity.  ``SmoothQuantModifier.mappings`` (upstream's ``[[balance patterns], smooth pattern]``
    pairs, or ``[smooth, [balance...]]`` / dicts as the AWQ resolver takes) restricts it to those."""
    from .smoothquant import ChannelMinMax, apply_smoothing, smoothquant_scales

    norms = {n: m for n, m in layer.named_modules()
             if "norm" in type(m).__name__.lower() and getattr(m, "weight", None) is not None and m.weight.dim() == 1}
    explicit = None
    if mappings:
        from .awq_module import _pattern_hits

        explicit = {}
        named = {n: m for n, m in layer.named_modules()}
        for mp in mappings:
            if isinstance(mp, dict):
                smooth_pat, bal_pats = mp["smooth_layer"] if "smooth_layer" in mp else mp["smooth_layers"], mp["balance_layers"]
            elif isinstance(mp[0], (list, tuple)):      # upstream SmoothQuant order: [[balance...], smooth]
                bal_pats, smooth_pat = mp[0], mp[1]
            else:
                smooth_pat, bal_pats = mp[0], mp[1]
            bal_pats = [bal_pats] if isinstance(bal_pats, str) else list(bal_pats)
            full = lambda n: f"{layer_name}.{n}" if layer_name and n else (layer_name or n)
            hits = [n for n in named if n and _pattern_hits(str(smooth_pat), full(n)) and getattr(named[n], "weight", None) is not None
                    and named[n].weight.dim() == 1]
            lins = [named[n] for n in named if isinstance(named[n], nn.Linear) and any(_pattern_hits(str(p), full(n)) for p in bal_pats)]
            if len(hits) == 1 and lins:
                explicit[hits[0]] = lins
        norms = {n: m for n, m in layer.named_modules() if n in explicit}
    if not norms:
        return
    stats: Dict[str, ChannelMinMax] = {}
    out_ref: Dict[str, torch.Tensor] = {}   # holds each norm output alive so its address is not reused
    consumers: Dict[str, List[nn.Linear]] = {n: [] for n in norms}

    def norm_hook(name):
        def fn(_m, _a, out):
            st = stats.setdefault(name, ChannelMinMax(out.shape[-1], dev))
            st.add(out.reshape(-1, out.shape[-1]))
            out_ref[name] = out
        return fn

    def lin_hook(mod):
        def fn(_m, a):
            x = a[0]
            for n, o in out_ref.items():
                same = (x.untyped_storage().data_ptr() == o.untyped_storage().data_ptr()
                        and x.storage_offset() == o.storage_offset() and x.shape == o.shape)
                if same and mod not in consumers[n]:
                    consumers[n].append(mod)
        return fn

    hooks = [m.register_forward_hook(norm_hook(n)) for n, m in norms.items()]
    hooks += [m.register_forward_pre_hook(lin_hook(m)) for m in layer.modules() if isinstance(m, nn.Linear)]
    for args, kwargs in cache:
        layer(*args, **kwargs)
    for hk in hooks:
        hk.remove()
    out_ref.clear()
    from .sharding import dist_world

    if dist_world()[0] > 1:
        import torch.distributed as dist

        from .sharding import allreduce_inplace

        for n in sorted(stats):       # the calibration samples are split over the ranks: global min / max
            allreduce_inplace(stats[n].cmin, dist.ReduceOp.MIN)
            allreduce_inplace(stats[n].cmax, dist.ReduceOp.MAX)
    for n, norm in norms.items():
        lins = explicit[n] if explicit is not None else consumers[n]
        if not lins or n not in stats:
            continue
        s = smoothquant_scales(stats[n], [l.weight.data for l in lins], alpha)
        vecs = [norm.weight.data] + ([norm.bias.data] if getattr(norm, "bias", None) is not None else [])
        new_w, new_v = apply_smoothing(s, [l.weight.data for l in lins], vecs)
        for l, w in zip(lins, new_w):
            l.weight.data.copy_(w)
        for v, nv in zip(vecs, new_v):
            v.copy_(nv)
