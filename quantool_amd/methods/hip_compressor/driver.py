"""Calibration-plugin driver of the MI355X backend.

One class, ``HipCompressorQuantizer``, stands where the reference has ``LLMCompressorQuantizer``
(``src/quantool/methods/llm_compressor/base.py:30-345``).  What a caller can observe is kept --
constructor state (``last_model``, ``last_output_dir``, ``last_tokenizer``, ``source_model``,
``_last_recipe``), how ``quantize(**kwargs)`` splits its keywords three ways, the defaults it
injects, the exception types, the returned path -- while the work behind ``oneshot`` is this
repo's HIP path.  The three method plugins are declared as data in ``plugins.py``.

Keyword routing of ``quantize`` (reference ``base.py:77-172``), as a table:

====================================  =========================================================
keyword                               goes to
====================================  =========================================================
``dataset=``                          ``oneshot_kwargs["dataset"]`` (overrides)
a name in ``oneshot``'s signature     ``oneshot_kwargs`` unless already set there
``method_kwargs__<key>``              ``method_kwargs[<key>]``
anything else                         dropped without a diagnostic
====================================  =========================================================
"""
from __future__ import annotations

import inspect
import logging
from pathlib import Path
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

from ...core.base import BaseQuantizer

RecipeType = Union[Any, List[Any]]

_CALIBRATION_KEYS = ("dataset", "dataset_path", "calibration_dataloader")
_TEXT_FALLBACK_COLUMNS = ("prompt", "completion", "chosen", "rejected", "label")
_METHOD_PREFIX = "method_kwargs__"


def _truthy(value) -> bool:
    """bool(value) where that is defined; otherwise "is not None" (tensors, lazy datasets)."""
    try:
        return bool(value)
    except Exception:  # noqa: BLE001
        return value is not None


class HipCompressorQuantizer(BaseQuantizer):
    #: names accepted by the engine's ``oneshot``; filled on first use, shared by all plugins
    _ONESHOT_PARAMS_CACHE: Optional[set] = None

    def __init__(self, model_id, *args, **kwargs):
        # cli.quantize_step builds plugins as create(method, model_id=..., **quantization_config)
        # (cli.py:201-203); the extra keywords are accepted and unused, as upstream.
        super().__init__(model_id)
        self.last_model = None
        self.last_tokenizer = None
        self.last_output_dir: Optional[Path] = None
        self.source_model = None
        self._last_recipe: Optional[RecipeType] = None

    # ---------------------------------------------------------------- engine lookup
    @staticmethod
    def _import_oneshot_static() -> Callable:
        from ...engine.oneshot import oneshot

        return oneshot

    def _import_oneshot(self) -> Callable:
        try:
            return self._import_oneshot_static()
        except ImportError as exc:
            raise ImportError("quantool_amd's HIP engine could not be imported; build the library first "
                              "(python -c 'import __graft_entry__ as g; g.build()')") from exc

    @classmethod
    def _get_oneshot_params(cls) -> set:
        """The engine's keyword names, by introspection (reference ``base.py:45-72``)."""
        if HipCompressorQuantizer._ONESHOT_PARAMS_CACHE is not None:
            return HipCompressorQuantizer._ONESHOT_PARAMS_CACHE
        names: set = set()
        try:
            sig = inspect.signature(cls._import_oneshot_static())
            names = {n for n in sig.parameters if n not in ("self", "unused")}
        except Exception as exc:  # noqa: BLE001 - routing degrades to "nothing is an engine kwarg"
            logging.getLogger(__name__).warning(f"oneshot signature unavailable ({exc}); no keyword is routed to it")
        HipCompressorQuantizer._ONESHOT_PARAMS_CACHE = names   # one cache for all plugins, on the base class
        return names

    # ---------------------------------------------------------------- calibration hooks
    def require_calibration(self) -> bool:
        return True

    def prepare_calibration_data(self, dataset, tokenizer=None):
        """Chat-template conversational rows (``utils/calibration_text.py``), then guarantee a
        ``text`` column, falling back to prompt/completion/chosen/rejected/label (reference
        ``base.py:271-315``).  Every failure is a warning: calibration data is passed on as is."""
        if tokenizer is not None:
            dataset = self._apply_chat_template(dataset, tokenizer)
        try:
            is_split_dict = hasattr(dataset, "keys") and not hasattr(dataset, "column_names")
            if is_split_dict:
                for split in list(dataset.keys()):
                    dataset[split] = self._with_text_column(dataset[split])
            else:
                dataset = self._with_text_column(dataset)
        except Exception as exc:  # noqa: BLE001
            self.logger.warning(f"could not normalise the calibration columns: {exc}")
        return dataset

    def _apply_chat_template(self, dataset, tokenizer):
        try:
            from ...utils.calibration_text import row_converter

            convert_row = row_converter()
            dataset = dataset.map(lambda row: convert_row(row, tokenizer), batched=False)
            self.logger.info("calibration rows rendered through the chat template")
        except Exception as exc:  # noqa: BLE001
            self.logger.warning(f"chat template not applied: {exc}")
        return dataset

    def _with_text_column(self, ds):
        columns = set(getattr(ds, "column_names", None) or ())
        if columns & {"text", "text_target"}:
            return ds
        source = next((c for c in _TEXT_FALLBACK_COLUMNS if c in columns), None)
        if source is None:
            return ds
        try:
            ds = ds.map(lambda row: {"text": row.get(source)}, batched=False)
            self.logger.info(f"'text' column derived from '{source}'")
        except Exception as exc:  # noqa: BLE001
            self.logger.warning(f"deriving 'text' from '{source}' failed: {exc}")
        return ds

    # ---------------------------------------------------------------- quantize
    def _build_recipe(self, level: Optional[str], method_kwargs: Dict[str, Any]) -> Tuple[RecipeType, str]:
        raise NotImplementedError

    def _split_keywords(self, loose: Dict[str, Any], oneshot_kwargs: Dict[str, Any],
                        method_kwargs: Dict[str, Any]) -> None:
        """Rows 2-4 of the routing table in the module docstring."""
        engine_names = self._get_oneshot_params()
        for key, value in loose.items():
            if key in engine_names:
                oneshot_kwargs.setdefault(key, value)
            elif key.startswith(_METHOD_PREFIX):
                method_kwargs[key[len(_METHOD_PREFIX):]] = value

    def _default_output_dir(self, level_hint: Optional[str]) -> Path:
        flat = lambda s: str(s).replace("/", "_")  # noqa: E731
        return Path("./output") / f"{self.name}_{flat(self.model_id) if self.model_id else 'model'}_{flat(level_hint or 'default')}"

    def _prepare_oneshot_kwargs(self, model, oneshot_kwargs: Dict[str, Any], level_hint: Optional[str]):
        """Engine defaults quantool relies on: compressed save, trusted remote code, an output
        directory that exists before the run (reference ``base.py:218-238``)."""
        out = {"model": model, "save_compressed": True, "trust_remote_code_model": True, **oneshot_kwargs}
        out["output_dir"] = str(out.get("output_dir") or self._default_output_dir(level_hint))
        Path(out["output_dir"]).mkdir(parents=True, exist_ok=True)
        return out

    def _has_calibration_source(self, oneshot_kwargs: Dict[str, Any]) -> bool:
        return any(_truthy(oneshot_kwargs.get(k)) for k in _CALIBRATION_KEYS)

    def quantize(self, model, level: Optional[str] = None, recipe: Optional[RecipeType] = None,
                 oneshot_kwargs: Optional[Dict[str, Any]] = None, method_kwargs: Optional[Dict[str, Any]] = None,
                 dataset: Optional[Any] = None, **kwargs) -> str:
        """Calibrate and quantize ``model``; returns the resolved output directory as a string."""
        self._reject_multiple_levels(level)
        engine_kw = dict(oneshot_kwargs or {})
        method_kw = dict(method_kwargs or {})
        if dataset is not None:
            engine_kw["dataset"] = dataset
        self._split_keywords(kwargs, engine_kw, method_kw)

        if recipe is None:
            recipe, level_hint = self._build_recipe(level, method_kw)
        else:
            level_hint = level or getattr(self, "default_level", "default")
        self._last_recipe = recipe

        engine = self._import_oneshot()
        engine_kw = self._prepare_oneshot_kwargs(model, engine_kw, level_hint)
        engine_kw.setdefault("recipe", recipe)
        if not self._has_calibration_source(engine_kw):
            raise ValueError("these quantizers require calibration data: pass `dataset`, or `dataset_path` / "
                             "`calibration_dataloader` inside `oneshot_kwargs`")

        self.source_model = model
        self.logger.info(f"oneshot -> {engine_kw['output_dir']}")
        try:
            self.last_model = engine(**engine_kw)
        except Exception as exc:
            self.logger.error(f"oneshot failed: {exc}")
            raise
        self.last_output_dir = Path(engine_kw["output_dir"]).resolve()
        self.logger.info(f"quantized model at {self.last_output_dir}")
        return str(self.last_output_dir)

    # ---------------------------------------------------------------- export hook
    def _save_model_files(self, save_directory) -> None:
        """Compressed save of the last result plus a tokenizer (reference ``base.py:174-216``)."""
        if not self.last_model:
            raise RuntimeError("No quantized model available: quantize() has not produced one yet")
        target = Path(save_directory)
        target.mkdir(parents=True, exist_ok=True)
        self.last_model.save_pretrained(str(target), save_compressed=True)
        if self.last_tokenizer is not None:
            self.last_tokenizer.save_pretrained(str(target))
            return
        try:
            from transformers import AutoTokenizer

            AutoTokenizer.from_pretrained(self.model_id, trust_remote_code=True).save_pretrained(str(target))
        except Exception as exc:  # noqa: BLE001 - a missing tokenizer does not fail the save upstream either
            self.logger.warning(f"tokenizer not saved: {exc}")
