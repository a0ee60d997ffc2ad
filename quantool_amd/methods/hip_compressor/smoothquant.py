"""``method=smoothquant`` on the MI355X backend (reference:
``src/quantool/methods/llm_compressor/smoothquant/smoothquant.py``)."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

from ...core.meta import TemplateQuantizationCard
from ...core.registry import QuantizerRegistry
from .base import HipCompressorQuantizer, RecipeType
from .gptq import check_scheme


@QuantizerRegistry.register
class SmoothQuant(HipCompressorQuantizer):
    """SmoothQuant pre-pass followed by GPTQ (two modifiers, smoothquant.py:77-84)."""

    name = "smoothquant"
    supported_levels = ["W8A8", "INT8", "W4A8"]
    template_card = TemplateQuantizationCard(
        title="SmoothQuant",
        description="Smoothing-based activation quantization for W8A8",
        hyperparameters={"method": "smoothquant", "scheme": "W8A8", "smoothing_strength": 0.5, "targets": "Linear",
                         "ignore": ["lm_head"], "num_calibration_samples": 512},
        intended_use="W8A8 quantization with activation smoothing for better accuracy",
        limitations="Requires calibration dataset; best for W8A8 schemes",
        citations=["https://arxiv.org/abs/2211.10438"],
    )

    def _build_recipe(self, level: Optional[str], method_kwargs: Dict[str, Any]) -> Tuple[RecipeType, str]:
        """Default scheme W8A8, smoothing_strength 0.5; the GPTQ stage gets scheme / targets /
        ignore only -- no block_size / dampening pass-through (smoothquant.py:62-84)."""
        from ...engine.modifiers import GPTQModifier, SmoothQuantModifier

        scheme = level or method_kwargs.get("scheme", "W8A8")
        check_scheme(scheme)
        smoothing_strength = method_kwargs.get("smoothing_strength", 0.5)
        recipe: List[Any] = [
            SmoothQuantModifier(smoothing_strength=smoothing_strength),
            GPTQModifier(scheme=scheme, targets=method_kwargs.get("targets", "Linear"),
                         ignore=method_kwargs.get("ignore", ["lm_head"])),
        ]
        self.logger.info(f"Built SmoothQuant recipe with scheme={scheme}, smoothing_strength={smoothing_strength}")
        return recipe, scheme
