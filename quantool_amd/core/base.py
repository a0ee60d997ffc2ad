"""Base class of every method plugin.

Reference surface: ``BaseQuantizer`` (``src/quantool/core/base.py:7-33``) with the hooks it
inherits from ``CalibrationMixin`` (``core/helpers/calibration_mixin.py:4-30``) and the local-save
half of ``ExportMixin`` (``core/helpers/export_mixin.py:17-81``).  Hub upload
(``export_mixin.py:83-139``) is the reference's control plane and stays there.
"""
from __future__ import annotations

import logging
import os
from abc import ABC, abstractmethod
from typing import List, Union

from .meta import TemplateQuantizationCard


class BaseQuantizer(ABC):
    name: str
    supported_levels: list
    supports_multiple_levels: bool = False
    template_card: TemplateQuantizationCard

    def __init__(self, model_id, *args, **kwargs):
        self.model_id = model_id
        self.logger = logging.getLogger(f"quantool_amd.{self.__class__.__name__}")

    # -- quantization -------------------------------------------------------------------------
    @abstractmethod
    def quantize(self, model, level: Union[str, List[str]], **kwargs) -> Union[str, List[str]]:
        """Apply quantization at the given level(s); returns the output path(s)."""

    def _reject_multiple_levels(self, level) -> None:
        if isinstance(level, list) and not self.supports_multiple_levels:
            raise ValueError(
                f"Method '{self.name}' does not support multiple quantization levels. "
                f"Please specify a single level.")

    # -- calibration hooks (defaults as CalibrationMixin) -----------------------------------------
    def require_calibration(self) -> bool:
        return False

    def prepare_calibration_data(self, dataset, tokenizer=None):
        return dataset

    def run_calibration(self):
        return None

    # -- export hooks (local half of ExportMixin) ---------------------------------------------------
    def _save_model_files(self, save_directory):
        raise NotImplementedError("Subclasses must implement _save_model_files method")

    def save_pretrained(self, save_directory) -> None:
        os.makedirs(save_directory, exist_ok=True)
        self.logger.info(f"Saving model files to {save_directory}")
        self._save_model_files(save_directory)

    def save_model_card(self, save_directory) -> None:
        os.makedirs(save_directory, exist_ok=True)
        card = getattr(self, "template_card", None)
        if card is None:
            self.logger.warning("No template_card attribute found, skipping model card generation")
            return
        path = os.path.join(save_directory, "README.md")
        with open(path, "w", encoding="utf-8") as fh:
            fh.write(card.to_markdown())
        self.logger.info(f"Model card saved to {path}")
