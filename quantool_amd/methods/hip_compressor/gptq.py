"""``method=gptq`` on the MI355X backend (reference: ``src/quantool/methods/llm_compressor/gptq/gptq.py``)."""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

from ...core.meta import TemplateQuantizationCard
from ...core.registry import QuantizerRegistry
from ...engine.schemes import is_preset_scheme
from .base import HipCompressorQuantizer, RecipeType

_VALID = ("W8A16, W4A16, W4A16_ASYM, W8A8, INT8, W4A8, FP8, FP8_DYNAMIC, FP8_BLOCK, NVFP4A16, NVFP4, "
          "UNQUANTIZED")


def check_scheme(scheme) -> None:
    if not is_preset_scheme(scheme):
        raise ValueError(f"Scheme '{scheme}' is not a valid compressed-tensors preset scheme. "
                         f"Valid schemes include: {_VALID}")


@QuantizerRegistry.register
class GPTQ(HipCompressorQuantizer):
    """GPTQ: Hessian-aware post-training weight quantization with error feedback."""

    name = "gptq"
    supported_levels = ["W4A16", "W8A8", "INT8", "W8A16", "W4A16_ASYM", "W4A8"]
    template_card = TemplateQuantizationCard(
        title="GPTQ Quantization",
        description="Post-training quantization using GPTQ algorithm with calibration data",
        hyperparameters={"method": "gptq", "scheme": "W4A16", "targets": "Linear", "ignore": ["lm_head"],
                         "num_calibration_samples": 512},
        intended_use="Efficient inference for LLMs with minimal accuracy loss",
        limitations="Requires calibration dataset; quantization time scales with model size",
        citations=["https://arxiv.org/abs/2210.17323"],
    )

    def _build_recipe(self, level: Optional[str], method_kwargs: Dict[str, Any]) -> Tuple[RecipeType, str]:
        """scheme = level or method_kwargs['scheme'] or W4A16; pass-through of block_size,
        dampening_frac, sequential_targets only (gptq.py:59-84)."""
        from ...engine.modifiers import GPTQModifier

        scheme = level or method_kwargs.get("scheme", "W4A16")
        check_scheme(scheme)
        if scheme not in self.supported_levels:
            self.logger.warning(f"Level '{scheme}' not in supported list, using anyway: {self.supported_levels}")
        modifier_kwargs = {"scheme": scheme, "targets": method_kwargs.get("targets", "Linear"),
                           "ignore": method_kwargs.get("ignore", ["lm_head"])}
        for key in ("block_size", "dampening_frac", "sequential_targets"):
            if key in method_kwargs:
                modifier_kwargs[key] = method_kwargs[key]
        recipe = GPTQModifier(**modifier_kwargs)
        self.logger.info(f"Built GPTQ recipe with scheme={scheme}, targets={modifier_kwargs['targets']}")
        return recipe, scheme
