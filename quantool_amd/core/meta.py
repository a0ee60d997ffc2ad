"""Model-card metadata carried by every method plugin (reference: ``src/quantool/core/meta.py:5-21``)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List


@dataclass
class TemplateQuantizationCard:
    title: str
    description: str
    hyperparameters: Dict[str, Any] = field(default_factory=dict)
    intended_use: str = ""
    limitations: str = ""
    citations: List[str] = field(default_factory=list)

    def to_markdown(self) -> str:
        lines = [f"# {self.title}", "", self.description, ""]
        if self.hyperparameters:
            lines += ["## Quantization hyperparameters", ""]
            lines += [f"- `{k}`: `{v}`" for k, v in self.hyperparameters.items()]
            lines.append("")
        if self.intended_use:
            lines += ["## Intended use", "", self.intended_use, ""]
        if self.limitations:
            lines += ["## Limitations", "", self.limitations, ""]
        if self.citations:
            lines += ["## Citations", ""] + [f"- {c}" for c in self.citations] + [""]
        return "\n".join(lines)
