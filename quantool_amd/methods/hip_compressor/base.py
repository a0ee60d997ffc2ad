"""Shared driver of the three calibration plugins.

Behavioural mirror of ``LLMCompressorQuantizer`` (``src/quantool/methods/llm_compressor/base.py:30-345``):
same constructor state, kwargs routing, defaults, errors and return value; the engine behind
``oneshot`` is this repo's HIP backend instead of llm-compressor.
"""
from __future__ import annotations

import inspect
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

from ...core.base import BaseQuantizer

RecipeType = Union[Any, List[Any]]


class HipCompressorQuantizer(BaseQuantizer):
    _ONESHOT_PARAMS_CACHE: Optional[set] = None

    def __init__(self, model_id, *args, **kwargs):
        # extra constructor kwargs (the CLI passes **quantization_config, cli.py:201-203) are ignored
        super().__init__(model_id)
        self.last_output_dir: Optional[Path] = None
        self.last_model = None
        self.last_tokenizer = None
        self.source_model = None
        self._last_recipe: Optional[RecipeType] = None

    # ------------------------------------------------------------------ public API
    @classmethod
    def _get_oneshot_params(cls) -> set:
        """Names ``oneshot`` accepts, read from its signature (base.py:45-72)."""
        if cls._ONESHOT_PARAMS_CACHE is None:
            try:
                fn = cls._import_oneshot_static()
                cls._ONESHOT_PARAMS_CACHE = {p for p in inspect.signature(fn).parameters if p not in ("self", "unused")}
            except Exception as exc:  # noqa: BLE001
                import logging

                logging.getLogger(__name__).warning(f"Could not extract oneshot parameters: {exc}. Using empty set.")
                cls._ONESHOT_PARAMS_CACHE = set()
        return cls._ONESHOT_PARAMS_CACHE

    def require_calibration(self):
        return True

    def quantize(self, model, level: Optional[str] = None, recipe: Optional[RecipeType] = None,
                 oneshot_kwargs: Optional[Dict[str, Any]] = None, method_kwargs: Optional[Dict[str, Any]] = None,
                 dataset: Optional[Any] = None, **kwargs) -> str:
        """Run the oneshot flow and return the output directory path (base.py:77-172)."""
        self._reject_multiple_levels(level)
        oneshot_kwargs = dict(oneshot_kwargs or {})
        method_kwargs = dict(method_kwargs or {})
        if dataset is not None:
            oneshot_kwargs["dataset"] = dataset

        valid = self._get_oneshot_params()
        for key in list(kwargs):
            if key in valid:                       # explicit oneshot_kwargs win (setdefault)
                oneshot_kwargs.setdefault(key, kwargs.pop(key))
        for key in list(kwargs):
            if key.startswith("method_kwargs__"):
                method_kwargs[key.split("__", 1)[1]] = kwargs.pop(key)
        # whatever is left in kwargs is silently ignored, as in the reference (base.py:126-130)

        if recipe is None:
            recipe, inferred_level = self._build_recipe(level, method_kwargs)
        else:
            inferred_level = level or getattr(self, "default_level", "default")
        self._last_recipe = recipe

        oneshot_fn = self._import_oneshot()
        oneshot_kwargs = self._prepare_oneshot_kwargs(model, oneshot_kwargs, inferred_level)
        oneshot_kwargs.setdefault("recipe", recipe)

        if not self._has_calibration_source(oneshot_kwargs):
            raise ValueError(
                "llm-compressor integrations require calibration data. "
                "Provide `dataset`, `dataset_path`, or a custom `calibration_dataloader` "
                "through `oneshot_kwargs`.")

        self.logger.info(f"Running oneshot with output_dir={oneshot_kwargs.get('output_dir')}")
        self.source_model = model
        try:
            self.last_model = oneshot_fn(**oneshot_kwargs)
        except Exception as exc:
            self.logger.error(f"oneshot failed: {exc}")
            raise
        self.last_output_dir = Path(oneshot_kwargs["output_dir"]).resolve()
        self.logger.info(f"Quantization complete. Model ready at: {self.last_output_dir}")
        return str(self.last_output_dir)

    # ------------------------------------------------------------------ export hook
    def _save_model_files(self, save_directory):
        if not self.last_model:
            raise RuntimeError("No quantized model available. Call `quantize()` before saving.")
        dest = Path(save_directory)
        dest.mkdir(parents=True, exist_ok=True)
        self.logger.info(f"Saving quantized model to: {dest}")
        self.last_model.save_pretrained(str(dest), save_compressed=True)
        if self.last_tokenizer is not None:
            self.last_tokenizer.save_pretrained(str(dest))
        else:
            try:
                from transformers import AutoTokenizer

                AutoTokenizer.from_pretrained(self.model_id, trust_remote_code=True).save_pretrained(str(dest))
            except Exception as exc:  # noqa: BLE001 - the reference downgrades this to a warning
                self.logger.warning(f"Could not save tokenizer: {exc}")

    # ------------------------------------------------------------------ internals
    def _build_recipe(self, level: Optional[str], method_kwargs: Dict[str, Any]) -> Tuple[RecipeType, str]:
        raise NotImplementedError

    def _default_output_dir(self, level_hint: Optional[str]) -> Path:
        model_name = str(self.model_id).replace("/", "_") if self.model_id else "model"
        level_fragment = (level_hint or "default").replace("/", "_")
        return Path("./output") / f"{self.name}_{model_name}_{level_fragment}"

    def _prepare_oneshot_kwargs(self, model, oneshot_kwargs: Dict[str, Any], level_hint: Optional[str]):
        prepared = dict(oneshot_kwargs)
        prepared.setdefault("model", model)
        prepared.setdefault("save_compressed", True)
        prepared.setdefault("trust_remote_code_model", True)
        output_dir = prepared.get("output_dir") or self._default_output_dir(level_hint)
        prepared["output_dir"] = str(output_dir)
        Path(prepared["output_dir"]).mkdir(parents=True, exist_ok=True)
        return prepared

    def _has_calibration_source(self, oneshot_kwargs: Dict[str, Any]) -> bool:
        def present(v):
            try:
                return bool(v)
            except Exception:  # tensors / datasets with ambiguous truth value
                return v is not None
        return any(present(oneshot_kwargs.get(k)) for k in ("dataset", "dataset_path", "calibration_dataloader"))

    @staticmethod
    def _import_oneshot_static():
        from ...engine.oneshot import oneshot

        return oneshot

    def _import_oneshot(self):
        try:
            return self._import_oneshot_static()
        except ImportError as exc:
            raise ImportError(
                "The quantool_amd HIP backend is required for this quantizer. Build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'`.") from exc

    def prepare_calibration_data(self, dataset, tokenizer=None):
        """Chat-template the rows when quantool's textifier is importable, then make sure a
        ``text`` column exists (fallbacks as base.py:271-315)."""
        def ensure_text(ds):
            cols = set(getattr(ds, "column_names", []) or [])
            if "text" in cols or "text_target" in cols:
                return ds
            fallback = next((c for c in ("prompt", "completion", "chosen", "rejected", "label") if c in cols), None)
            if fallback is None:
                return ds
            try:
                ds = ds.map(lambda ex: {"text": ex.get(fallback)}, batched=False)
                self.logger.info(f"Created 'text' column from fallback '{fallback}' for oneshot")
            except Exception as exc:  # noqa: BLE001
                self.logger.warning(f"Failed to create 'text' fallback column from '{fallback}': {exc}")
            return ds

        if tokenizer is not None:
            try:
                from quantool.utils.dataset_textifier import convert_row  # quantool's own front-end (N3)

                dataset = dataset.map(lambda ex: convert_row(ex, tokenizer), batched=False)
                self.logger.info("Applied chat template processing to calibration dataset")
            except Exception as exc:  # noqa: BLE001
                self.logger.warning(f"Failed to apply chat template processing: {exc}")
        try:
            if hasattr(dataset, "keys") and not hasattr(dataset, "column_names"):
                for split in list(dataset.keys()):
                    dataset[split] = ensure_text(dataset[split])
            else:
                dataset = ensure_text(dataset)
        except Exception as exc:  # noqa: BLE001
            self.logger.warning(f"Error while ensuring text column for calibration dataset: {exc}")
        return dataset
