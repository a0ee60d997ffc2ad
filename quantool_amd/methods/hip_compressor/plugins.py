"""``method=gptq | awq | smoothquant`` on the MI355X backend, declared as data.

The reference spells each method out as a class of its own
(``src/quantool/methods/llm_compressor/{gptq/gptq.py:12-91, awq/awq.py:12-84,
smoothquant/smoothquant.py:12-90}``); they differ only in the facts tabulated in ``METHODS`` below,
so here one factory turns each row into a registered ``HipCompressorQuantizer`` subclass.

Observable behaviour kept per method (checked by ``tests/test_boundary.py``):

* registry name and ``supported_levels`` (order included);
* scheme resolution: ``level``, else ``method_kwargs["scheme"]``, else the row's default;
* a scheme that is not a compressed-tensors preset -> ``ValueError``; a preset outside
  ``supported_levels`` -> a warning only;
* ``targets`` / ``ignore`` default to ``"Linear"`` / ``["lm_head"]`` and are read from
  ``method_kwargs`` only; of the remaining ``method_kwargs`` just the row's ``forwarded`` keys
  reach the modifier (so e.g. ``actorder`` cannot be set from a quantool YAML);
* SmoothQuant is a two-stage recipe whose GPTQ stage receives no forwarded keys.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Tuple

from ...core.meta import TemplateQuantizationCard
from ...core.registry import QuantizerRegistry
from ...engine import modifiers as _mod
from ...engine.schemes import PRESET_SCHEMES, is_preset_scheme
from .driver import HipCompressorQuantizer, RecipeType


def check_scheme(scheme) -> None:
    """``ValueError`` unless ``scheme`` names a compressed-tensors preset (gptq.py:67-72)."""
    if is_preset_scheme(scheme):
        return
    known = ", ".join(sorted(PRESET_SCHEMES))
    raise ValueError(f"'{scheme}' is not a valid compressed-tensors preset scheme (known presets: {known})")


def _single(modifier_cls) -> Callable[[Dict[str, Any], Dict[str, Any]], RecipeType]:
    return lambda common, extra: modifier_cls(**common, **extra)


def _smooth_then_gptq(common: Dict[str, Any], extra: Dict[str, Any]) -> RecipeType:
    strength = extra.get("smoothing_strength", 0.5)
    return [_mod.SmoothQuantModifier(smoothing_strength=strength), _mod.GPTQModifier(**common)]


@dataclass(frozen=True)
class MethodSpec:
    name: str
    class_name: str
    levels: Tuple[str, ...]
    default_scheme: str
    forwarded: Tuple[str, ...]          # method_kwargs keys handed to ``assemble`` besides targets / ignore
    assemble: Callable[[Dict[str, Any], Dict[str, Any]], RecipeType]
    off_list_warning: str               # {scheme}, {levels} are substituted
    title: str
    summary: str
    use: str
    caveats: str
    paper: str
    card_extra: Tuple[Tuple[str, Any], ...] = ()


METHODS: Tuple[MethodSpec, ...] = (
    MethodSpec(
        name="gptq", class_name="GPTQ",
        levels=("W4A16", "W8A8", "INT8", "W8A16", "W4A16_ASYM", "W4A8"),
        default_scheme="W4A16",
        forwarded=("block_size", "dampening_frac", "sequential_targets"),
        assemble=_single(_mod.GPTQModifier),
        off_list_warning="scheme '{scheme}' is outside GPTQ's listed levels {levels}; continuing with it",
        title="GPTQ Quantization",
        summary="Second-order post-training weight quantization: each Linear is rounded column by column "
                "with the error fed forward through the inverse Hessian of its calibration inputs. "
                "Calibrated on AMD MI355X by quantool_amd's HIP kernels.",
        use="Serving LLMs from 4- or 8-bit weights at close to the original accuracy.",
        caveats="Needs a calibration set; one Hessian factorisation per Linear, so wall time grows with "
                "hidden size cubed.",
        paper="https://arxiv.org/abs/2210.17323",
    ),
    MethodSpec(
        name="awq", class_name="AWQ",
        levels=("W4A16", "W4A16_ASYM", "W8A16"),
        default_scheme="W4A16",
        forwarded=("mappings", "smoothing_strength"),
        assemble=_single(_mod.AWQModifier),
        off_list_warning="AWQ only supports weight-only schemes with 16-bit activations; '{scheme}' is not one "
                         "of {levels} and may not work",
        title="AWQ Quantization",
        summary="Activation-aware weight quantization: input channels that carry large activations are "
                "scaled up before rounding (scale found by a 20-point grid search per layer group), then "
                "weights are quantized round-to-nearest. Calibrated on AMD MI355X by quantool_amd's HIP kernels.",
        use="Weight-only 4-bit deployment where plain round-to-nearest loses too much accuracy.",
        caveats="Needs a calibration set; activations stay in 16-bit.",
        paper="https://arxiv.org/abs/2306.00978",
    ),
    MethodSpec(
        name="smoothquant", class_name="SmoothQuant",
        levels=("W8A8", "INT8", "W4A8"),
        default_scheme="W8A8",
        forwarded=("smoothing_strength",),
        assemble=_smooth_then_gptq,
        off_list_warning="",                                  # upstream does not check the list for this method
        title="SmoothQuant",
        summary="Migrates activation outliers into the weights with per-channel scales "
                "s = amax(X)^a / amax(W)^(1-a), then quantizes weights with GPTQ for W8A8 execution. "
                "Calibrated on AMD MI355X by quantool_amd's HIP kernels.",
        use="INT8 weight-and-activation inference.",
        caveats="Needs a calibration set; designed around 8-bit activations.",
        paper="https://arxiv.org/abs/2211.10438",
        card_extra=(("smoothing_strength", 0.5),),
    ),
)


def _card(spec: MethodSpec) -> TemplateQuantizationCard:
    hp: Dict[str, Any] = {"method": spec.name, "scheme": spec.default_scheme}
    hp.update(spec.card_extra)
    hp.update(targets="Linear", ignore=["lm_head"], num_calibration_samples=512)
    return TemplateQuantizationCard(title=spec.title, description=spec.summary, hyperparameters=hp,
                                    intended_use=spec.use, limitations=spec.caveats, citations=[spec.paper])


def _make_plugin(spec: MethodSpec) -> type:
    def _build_recipe(self, level, method_kwargs) -> Tuple[RecipeType, str]:
        scheme = level or method_kwargs.get("scheme", spec.default_scheme)
        check_scheme(scheme)
        if spec.off_list_warning and scheme not in self.supported_levels:
            self.logger.warning(spec.off_list_warning.format(scheme=scheme, levels=self.supported_levels))
        common = {"scheme": scheme, "targets": method_kwargs.get("targets", "Linear"),
                  "ignore": method_kwargs.get("ignore", ["lm_head"])}
        extra = {k: method_kwargs[k] for k in spec.forwarded if k in method_kwargs}
        recipe = spec.assemble(common, extra)
        self.logger.info(f"{spec.name} recipe: scheme={scheme} targets={common['targets']} extra={sorted(extra)}")
        return recipe, scheme

    cls = type(spec.class_name, (HipCompressorQuantizer,), {
        "__doc__": f"``method={spec.name}``: {spec.summary}",
        "__module__": __name__,
        "name": spec.name,
        "supported_levels": list(spec.levels),
        "template_card": _card(spec),
        "spec": spec,
        "_build_recipe": _build_recipe,
    })
    return QuantizerRegistry.register(cls)


PLUGINS: Dict[str, type] = {spec.name: _make_plugin(spec) for spec in METHODS}
GPTQ, AWQ, SmoothQuant = (PLUGINS[n] for n in ("gptq", "awq", "smoothquant"))

__all__: List[str] = ["GPTQ", "AWQ", "SmoothQuant", "PLUGINS", "METHODS", "MethodSpec", "check_scheme"]
