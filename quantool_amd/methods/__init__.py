"""Method plugins of the MI355X backend.

Importing the package walks its sub-packages so that their plugin classes land in
``QuantizerRegistry``; a backend that cannot be imported is reported through ``logging`` and
skipped, the behaviour quantool's own ``methods`` package has (``src/quantool/methods/__init__.py:9-14``:
with llm-compressor missing the registry simply lacks its three methods)."""
import logging
from importlib import import_module
from pkgutil import iter_modules
from typing import Dict, Optional

_log = logging.getLogger(__name__)

#: sub-package name -> None when it imported, else the exception that stopped it
BACKENDS: Dict[str, Optional[BaseException]] = {}


def _discover() -> None:
    for info in iter_modules(__path__):
        try:
            import_module(f"{__name__}.{info.name}")
        except Exception as exc:  # noqa: BLE001 - one broken backend must not empty the registry
            BACKENDS[info.name] = exc
            _log.error(f"method backend '{info.name}' not loaded: {exc}")
        else:
            BACKENDS[info.name] = None
            _log.info(f"method backend '{info.name}' loaded")


_discover()
