"""Plugin registry: method name -> plugin class.

Stands where ``src/quantool/core/registry.py:4-25`` does and answers the same four calls with the
same outcomes:

* ``register(cls)`` -- usable as a class decorator; ``ValueError`` when ``cls`` has no ``name``,
  ``KeyError`` when the name is taken.  The second rule is why this backend's ``gptq`` / ``awq`` /
  ``smoothquant`` replace the llm-compressor-backed plugins instead of sitting next to them.
* ``create(name, **kwargs)`` -- ``cls(**kwargs)``; an unknown name is a ``KeyError``.
* ``get(name)`` -- the class itself.
* ``list()`` -- names, oldest registration first.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Iterator, List


class Registry:
    def __init__(self) -> None:
        self._by_name: "OrderedDict[str, type]" = OrderedDict()

    def register(self, plugin_cls: type) -> type:
        key = getattr(plugin_cls, "name", None)
        if key is None:
            raise ValueError(f"{plugin_cls.__name__} must have a 'name' attribute to be registered")
        if key in self._by_name:
            raise KeyError(f"a plugin named {key!r} is already registered ({self._by_name[key].__qualname__})")
        self._by_name[key] = plugin_cls
        return plugin_cls

    def get(self, name: str) -> type:
        return self._by_name[name]

    def create(self, name: str, **kwargs):
        return self.get(name)(**kwargs)

    def list(self) -> List[str]:
        return [*self._by_name]

    def __contains__(self, name: object) -> bool:
        return name in self._by_name

    def __iter__(self) -> Iterator[str]:
        return iter(self._by_name)

    def __len__(self) -> int:
        return len(self._by_name)


QuantizerRegistry = Registry()
