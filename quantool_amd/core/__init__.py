"""Plugin contract of the MI355X backend: the same surface quantool's ``core`` package gives its
method plugins (``src/quantool/core/{base,registry,meta}.py``)."""
from .base import BaseQuantizer
from .meta import TemplateQuantizationCard
from .registry import QuantizerRegistry, Registry

__all__ = ["BaseQuantizer", "TemplateQuantizationCard", "QuantizerRegistry", "Registry"]
