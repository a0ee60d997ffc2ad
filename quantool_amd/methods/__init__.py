"""Method plugins.  As in the reference (``src/quantool/methods/__init__.py:9-14``) importing
this package imports every sub-package, whose classes register themselves; an import failure
is logged and swallowed so one broken backend does not take the registry down."""
import importlib
import logging
import pkgutil

logger = logging.getLogger(__name__)

for _finder, _name, _ispkg in pkgutil.iter_modules(__path__):
    try:
        importlib.import_module(f"{__name__}.{_name}")
        logger.info(f"Imported module: {_name}")
    except Exception as exc:  # noqa: BLE001 - mirror of the reference's catch-all
        logger.error(f"Failed to import module {_name}: {exc}")
