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
import random
import types
from pathlib import Path
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn

from .gptq_linear import HessianAccumulator, gptq_quantize_shared
from .modifiers import AWQModifier, GPTQModifier, SmoothQuantModifier
from .streams import GroupStreams

logger = logging.getLogger(__name__)


class _StopForward(Exception):
    pass


def find_decoder_layers(model: nn.Module) -> nn.ModuleList:
    """The longest ModuleList of identical-class blocks (model.model.layers, model.model.decoder.layers, ...)."""
    best = None
    for _, m in model.named_modules():
        if isinstance(m, nn.ModuleList) and len(m) > 0 and len({type(x) for x in m}) == 1:
            if best is None or len(m) > len(best):
                best = m
    if best is None:
        raise ValueError("could not locate the decoder layers (no homogeneous nn.ModuleList found)")
    return best


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


def _to_dev(x, dev):
    if isinstance(x, torch.Tensor):
        return x.to(dev)
    if isinstance(x, (list, tuple)):
        return type(x)(_to_dev(v, dev) for v in x)
    if isinstance(x, dict):
        return {k: _to_dev(v, dev) for k, v in x.items()}
    return x


def _save_compressed(model: nn.Module, save_directory, save_compressed: bool = True, **_):
    from .serialization import quantization_config, result_tensors, save_state

    results: Dict[str, Any] = getattr(model, "_qt_results", {})
    meta = getattr(model, "_qt_meta", {})
    state: Dict[str, torch.Tensor] = {}
    quantized = set(results)
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
    save_state(state, quantization_config(meta.get("weights", {}), meta.get("format", "pack-quantized"),
                                          meta.get("ignore", []), meta.get("input_activations")), save_directory, base_cfg)


def oneshot_module(model, dataset, recipe, dev, *, num_calibration_samples: int, max_seq_length: int, shuffle: bool,
                   tokenizer=None, dataloader=None, dataset_path=None, text_column: str = "text",
                   trust_remote_code: bool = False, seed: int = 42):
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
        model = AutoModelForCausalLM.from_pretrained(path, torch_dtype=torch.bfloat16,
                                                     trust_remote_code=trust_remote_code, local_files_only=True)
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
    model.eval()
    model.to(dev)
    layers = find_decoder_layers(model)
    qargs = qm.weight_args()

    # ---- inputs of the first decoder layer -------------------------------------------------------
    cache: List[tuple] = []

    def grab(_mod, args, kwargs):
        cache.append((args, kwargs))
        raise _StopForward

    h = layers[0].register_forward_pre_hook(grab, with_kwargs=True)
    with torch.no_grad():
        for b in batches:
            try:
                model(**_to_dev(b, dev), use_cache=False)
            except _StopForward:
                pass
    h.remove()

    prefix_of = {id(m): n for n, m in model.named_modules()}
    results: Dict[str, Any] = {}
    with torch.no_grad():
        for li, layer in enumerate(layers):
            lname = prefix_of[id(layer)]
            if aw is not None:
                from .awq_module import awq_layer

                results.update(awq_layer(layer, lname, cache, aw, dev))
                cache = _advance(layer, cache)
                continue
            linears = {f"{lname}.{n}" if n else lname: m for n, m in layer.named_modules()
                       if isinstance(m, nn.Linear) and gp.wants(f"{lname}.{n}", m)}
            if sq is not None:
                _smooth_layer(layer, cache, sq.smoothing_strength, dev)
            # discovery pass on batch 0: which Linears read the same tensor.  The input tensors are
            # kept alive until the grouping is done: a freed activation's address can be handed to
            # a later, unrelated tensor of the same shape, and pointer equality would then lie.
            seen: Dict[str, torch.Tensor] = {}
            hooks = [m.register_forward_pre_hook((lambda name: lambda _m, a: seen.__setitem__(name, a[0]))(n))
                     for n, m in linears.items()]
            args, kwargs = cache[0]
            layer(*args, **kwargs)
            for hk in hooks:
                hk.remove()
            groups: Dict[tuple, List[str]] = {}
            for n in linears:
                t = seen[n]
                key = (t.untyped_storage().data_ptr(), t.storage_offset(), tuple(t.shape), tuple(t.stride()))
                groups.setdefault(key, []).append(n)
            seen.clear()
            leaders = {names[0]: names for names in groups.values()}
            accs = {lead: HessianAccumulator(linears[lead].in_features, dev) for lead in leaders}
            hooks = [linears[lead].register_forward_pre_hook(
                (lambda lead: lambda _m, a: accs[lead].add(a[0].reshape(-1, a[0].shape[-2], a[0].shape[-1])
                                                           if a[0].dim() >= 3 else a[0].unsqueeze(0)))(lead))
                for lead in leaders]
            for args, kwargs in cache:
                layer(*args, **kwargs)
            for hk in hooks:
                hk.remove()
            # one stream per input group, largest in_features first (longest chain): see streams.py
            pool = GroupStreams(dev)

            def quantize_group(lead, names):
                ws = [linears[n].weight.data for n in names]
                res = gptq_quantize_shared(ws, accs[lead], qargs, block_size=gp.block_size,
                                           dampening_frac=gp.dampening_frac)
                for n, r in zip(names, res):
                    linears[n].weight.data.copy_(r.dequantized(linears[n].weight.dtype))
                    results[n] = r

            for lead, names in sorted(leaders.items(), key=lambda kv: -linears[kv[0]].in_features):
                pool.run(lambda lead=lead, names=names: quantize_group(lead, names))
            pool.join()
            accs.clear()
            cache = _advance(layer, cache)
            logger.info(f"quantized {lname}: {len(linears)} Linears in {len(leaders)} input groups")
    model._qt_results = results
    acts = qm.resolved_scheme.input_activations
    model._qt_meta = {"weights": qargs.to_config(), "format": qm.resolved_scheme.format, "ignore": list(qm.ignore),
                      "input_activations": acts.to_config() if acts is not None else None}
    model.save_pretrained = types.MethodType(_save_compressed, model)
    return model


def _advance(layer: nn.Module, cache):
    """Next layer's inputs: this layer re-run on its cached inputs with the quantised weights."""
    new_cache = []
    for args, kwargs in cache:
        out = layer(*args, **kwargs)
        out = out[0] if isinstance(out, (tuple, list)) else out
        new_cache.append(((out,) + tuple(args[1:]), kwargs))
    return new_cache


def _smooth_layer(layer: nn.Module, cache, alpha: float, dev) -> None:
    """SmoothQuant pre-pass with the default decoder mappings: each norm followed by
    Linears that read its output ({q,k,v} <- input norm, {gate,up} / fc1 <- post-attention norm)."""
    from .smoothquant import ChannelMinMax, apply_smoothing, smoothquant_scales

    norms = {n: m for n, m in layer.named_modules()
             if "norm" in type(m).__name__.lower() and getattr(m, "weight", None) is not None and m.weight.dim() == 1}
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
    for n, norm in norms.items():
        lins = consumers[n]
        if not lins or n not in stats:
            continue
        s = smoothquant_scales(stats[n], [l.weight.data for l in lins], alpha)
        vecs = [norm.weight.data] + ([norm.bias.data] if getattr(norm, "bias", None) is not None else [])
        new_w, new_v = apply_smoothing(s, [l.weight.data for l in lins], vecs)
        for l, w in zip(lins, new_w):
            l.weight.data.copy_(w)
        for v, nv in zip(vecs, new_v):
            v.copy_(nv)
