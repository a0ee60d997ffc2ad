"""Name -> plugin-class registry (reference: ``src/quantool/core/registry.py:4-25``).

Semantics kept: a plugin class must carry a ``name``; registering a second class under a taken
name raises ``KeyError`` (so this backend's ``gptq`` / ``awq`` / ``smoothquant`` replace the
llm-compressor-backed ones, they cannot co-register -- SURVEY.md 8b); ``create`` forwards kwargs
to the constructor; ``list`` returns the names in registration order.
"""
from __future__ import annotations

from typing import Dict, List


class Registry:
    def __init__(self) -> None:
        self._plugins: Dict[str, type] = {}

    def register(self, plugin_cls: type) -> type:
        name = getattr(plugin_cls, "name", None)
        if name is None:
            raise ValueError(f"{plugin_cls.__name__} must have a 'name' attribute")
        if name in self._plugins:
            raise KeyError(f"Plugin {name!r} already registered")
        self._plugins[name] = plugin_cls
        return plugin_cls

    def create(self, name: str, **kwargs):
        return self._plugins[name](**kwargs)

    def list(self) -> List[str]:
        return list(self._plugins)

    def get(self, name: str) -> type:
        return self._plugins[name]


QuantizerRegistry = Registry()
