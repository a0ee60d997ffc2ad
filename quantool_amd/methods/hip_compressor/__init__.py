"""Calibration plugins backed by the MI355X HIP path.  Importing this package registers ``gptq``,
``awq`` and ``smoothquant`` -- the names the reference's ``quantool.methods.llm_compressor`` package
registers, so one of the two packages is installed, never both."""
from .driver import HipCompressorQuantizer
from .plugins import AWQ, GPTQ, METHODS, PLUGINS, SmoothQuant

__all__ = ["HipCompressorQuantizer", "GPTQ", "AWQ", "SmoothQuant", "PLUGINS", "METHODS"]
