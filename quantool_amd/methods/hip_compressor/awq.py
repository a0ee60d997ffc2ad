"""``method=awq`` on the MI355X backend (reference: ``src/quantool/methods/llm_compressor/awq/awq.py``)."""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

from ...core.meta import TemplateQuantizationCard
from ...core.registry import QuantizerRegistry
from .base import HipCompressorQuantizer, RecipeType
from .gptq import check_scheme


@QuantizerRegistry.register
class AWQ(HipCompressorQuantizer):
    """AWQ: activation-aware per-channel scaling before weight-only quantization."""

    name = "awq"
    supported_levels = ["W4A16", "W4A16_ASYM", "W8A16"]
    template_card = TemplateQuantizationCard(
        title="AWQ Quantization",
        description="Activation-aware weight quantization preserving salient weights",
        hyperparameters={"method": "awq", "scheme": "W4A16", "targets": "Linear", "ignore": ["lm_head"],
                         "num_calibration_samples": 512},
        intended_use="Weight-only quantization with better accuracy than naive PTQ",
        limitations="Requires calibration dataset; weight-only (activations remain fp16)",
        citations=["https://arxiv.org/abs/2306.00978"],
    )

    def _build_recipe(self, level: Optional[str], method_kwargs: Dict[str, Any]) -> Tuple[RecipeType, str]:
        """Pass-through of mappings and smoothing_strength only (awq.py:53-79)."""
        from ...engine.modifiers import AWQModifier

        scheme = level or method_kwargs.get("scheme", "W4A16")
        check_scheme(scheme)
        if scheme not in self.supported_levels:
            self.logger.warning("AWQ only supports weight-only quantization with 16-bit activations. "
                                f"Scheme '{scheme}' may not be compatible. Supported: {self.supported_levels}")
        modifier_kwargs = {"scheme": scheme, "targets": method_kwargs.get("targets", "Linear"),
                           "ignore": method_kwargs.get("ignore", ["lm_head"])}
        for key in ("mappings", "smoothing_strength"):
            if key in method_kwargs:
                modifier_kwargs[key] = method_kwargs[key]
        recipe = AWQModifier(**modifier_kwargs)
        self.logger.info(f"Built AWQ recipe with scheme={scheme}")
        return recipe, scheme
