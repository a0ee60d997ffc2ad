"""GPTQ / AWQ / SmoothQuant plugins backed by the MI355X HIP path (same registry names as the
reference's ``quantool.methods.llm_compressor`` package)."""
from .awq import AWQ
from .gptq import GPTQ
from .smoothquant import SmoothQuant

__all__ = ["GPTQ", "AWQ", "SmoothQuant"]
