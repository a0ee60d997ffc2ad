#!/usr/bin/env python3
"""Generates ``boundary_reference.json``: what the REFERENCE's own plumbing does with the probe
inputs of ``boundary_cases.py``.  Runs in the build container only (``/root/reference`` does not exist
on the GPU box; only the JSON travels).

The reference's plugin classes cannot be imported as they are: ``loguru``, ``llmcompressor`` and
``compressed_tensors`` are not installed -- ordinary import errors.  The arithmetic lives
in those packages and is NOT what is captured here; what is captured is the reference's OWN code on the
way to and from them, which only needs something importable under those names:

  * ``loguru``              -> a logger object that counts ``warning`` / ``error`` calls;
  * ``llmcompressor``       -> ``oneshot`` = the recording function of ``boundary_cases.recording_oneshot``;
                               ``GPTQModifier`` / ``AWQModifier`` / ``SmoothQuantModifier`` = classes that keep
                               the keyword arguments they were built with;
  * ``compressed_tensors``  -> ``is_preset_scheme`` over the preset list the reference itself prints
                               (``gptq.py:68-70``).

Reference code exercised: ``quantool.core.registry`` (``registry.py:4-25``), ``LLMCompressorQuantizer``
(``llm_compressor/base.py:77-172,217-255,257-345``), ``GPTQ/AWQ/SmoothQuant._build_recipe``
(``gptq.py:46-91``, ``awq.py:41-84``, ``smoothquant.py:49-90``), ``convert_row`` and helpers
(``utils/dataset_textifier.py:15-260``).

usage:  python tests/golden/make_boundary_fixtures.py        (rewrites tests/golden/boundary_reference.json)
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import tempfile
import textwrap
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF_SRC = Path("/root/reference/src")
OUT = HERE / "boundary_reference.json"

sys.path.insert(0, str(HERE))
import boundary_cases as bc  # noqa: E402

STUBS = {
    "loguru/__init__.py": """
        COUNTS = {"warning": 0, "error": 0}

        class _Logger:
            def bind(self, **kw): return self
            def opt(self, *a, **kw): return self
            def remove(self, *a, **kw): pass
            def add(self, *a, **kw): return 0
            def configure(self, *a, **kw): pass
            def level(self, *a, **kw): pass
            def debug(self, *a, **kw): pass
            def info(self, *a, **kw): pass
            def success(self, *a, **kw): pass
            def exception(self, *a, **kw): COUNTS["error"] += 1
            def critical(self, *a, **kw): COUNTS["error"] += 1
            def warning(self, *a, **kw): COUNTS["warning"] += 1
            def error(self, *a, **kw): COUNTS["error"] += 1

        logger = _Logger()
    """,
    "llmcompressor/__init__.py": """
        from loguru import logger
        CURRENT = {"oneshot": None}

        def _dispatch(**kw):
            return CURRENT["oneshot"](**kw)

        def __getattr__(name):
            if name == "oneshot":          # whatever the generator installed for the current probe
                return CURRENT["oneshot"]
            raise AttributeError(name)
    """,
    "llmcompressor/modifiers/__init__.py": "",
    "llmcompressor/modifiers/_recording.py": """
        class Recording:
            def __init__(self, **kwargs):
                self.kwargs = dict(kwargs)
    """,
    "llmcompressor/modifiers/quantization/__init__.py": """
        from .._recording import Recording
        class GPTQModifier(Recording): pass
    """,
    "llmcompressor/modifiers/awq/__init__.py": """
        from .._recording import Recording
        class AWQModifier(Recording): pass
    """,
    "llmcompressor/modifiers/smoothquant/__init__.py": """
        from .._recording import Recording
        class SmoothQuantModifier(Recording): pass
    """,
    "compressed_tensors/__init__.py": "",
    "compressed_tensors/quantization/__init__.py": """
        PRESETS = %r

        def is_preset_scheme(name):
            return isinstance(name, str) and name.upper() in PRESETS
    """ % (bc.PRESETS,),
}


def install_stubs() -> Path:
    root = Path(tempfile.mkdtemp(prefix="qt_boundary_stubs_"))
    for rel, body in STUBS.items():
        p = root / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(textwrap.dedent(body))
    sys.path.insert(0, str(root))
    return root


def canon_recipe(r):
    if isinstance(r, (list, tuple)):
        return [canon_recipe(x) for x in r]
    if hasattr(r, "kwargs") and type(r).__name__.endswith("Modifier"):
        return {"cls": type(r).__name__, "kwargs": bc.jsonable(r.kwargs)}
    return bc.jsonable(r)


def main():
    if not REF_SRC.is_dir():
        raise SystemExit(f"{REF_SRC} not found: this generator runs in the build container only")
    install_stubs()
    sys.path.insert(0, str(REF_SRC))
    scratch = Path(tempfile.mkdtemp(prefix="qt_boundary_run_"))
    os.chdir(scratch)                       # the reference creates ./output/... and (with real loguru) ./logs

    import loguru
    import llmcompressor
    import quantool.methods  # noqa: F401  (auto-imports the sub-packages, methods/__init__.py:9-14)
    from quantool.core.registry import QuantizerRegistry, Registry
    from quantool.core.base import BaseQuantizer
    from quantool.utils import dataset_textifier as dt

    out = {"oneshot_params": bc.ONESHOT_PARAMS, "presets": bc.PRESETS}

    # ---- registry --------------------------------------------------------------------------------
    reg = {"listed": sorted(QuantizerRegistry.list())}
    r = Registry()

    class NoName:
        pass

    try:
        r.register(NoName)
    except Exception as e:  # noqa: BLE001
        reg["register_without_name"] = type(e).__name__

    class A(BaseQuantizer):
        name = "a"
        supported_levels = []

        def quantize(self, model, level, **kw):
            return "x"

    reg["register_returns_class"] = r.register(A) is A
    try:
        r.register(A)
    except Exception as e:  # noqa: BLE001
        reg["register_twice"] = type(e).__name__
    inst = r.create("a", model_id="m/n")
    reg["create_sets_model_id"] = inst.model_id
    reg["list_after"] = r.list()
    try:
        r.create("missing")
    except Exception as e:  # noqa: BLE001
        reg["create_missing"] = type(e).__name__
    out["registry"] = reg

    # ---- class attributes -------------------------------------------------------------------------
    attrs = {}
    for m in ("gptq", "awq", "smoothquant"):
        q = QuantizerRegistry.create(m, model_id="org/model", targets="Linear", ignore=["lm_head"])
        card = q.template_card
        attrs[m] = {
            "class": type(q).__name__, "name": q.name, "supported_levels": list(q.supported_levels),
            "supports_multiple_levels": q.supports_multiple_levels, "require_calibration": q.require_calibration(),
            "card_title": card.title, "card_hyperparameters": bc.jsonable(card.hyperparameters),
            "card_citations": list(card.citations),
            "initial_state": {k: bc.jsonable(getattr(q, k)) for k in
                              ("last_output_dir", "last_model", "last_tokenizer", "source_model", "_last_recipe")},
            "model_id": q.model_id,
        }
    out["class_attrs"] = attrs

    # ---- _build_recipe ----------------------------------------------------------------------------
    recipes = {}
    for cid, m, level, mk in bc.RECIPE_CASES:
        q = QuantizerRegistry.create(m, model_id="m")
        loguru.COUNTS.update(warning=0, error=0)
        try:
            recipe, scheme = q._build_recipe(level, dict(mk))
            recipes[cid] = {"scheme": scheme, "recipe": canon_recipe(recipe), "warnings": loguru.COUNTS["warning"]}
        except Exception as e:  # noqa: BLE001
            recipes[cid] = {"raises": type(e).__name__}
    out["recipes"] = recipes

    # ---- quantize() ---------------------------------------------------------------------------------
    DATASET, RECIPE = object(), object()

    def subst(x, tmp):
        if isinstance(x, dict):
            return {k: subst(v, tmp) for k, v in x.items()}
        if isinstance(x, list):
            return [subst(v, tmp) for v in x]
        if x == "@DATASET":
            return DATASET
        if x == "@RECIPE":
            return RECIPE
        if isinstance(x, str) and x.startswith("@TMP"):
            return str(tmp) + x[4:]
        return x

    def unsubst(x, tmp):
        if x is DATASET:
            return "@DATASET"
        if x is RECIPE:
            return "@RECIPE"
        if isinstance(x, dict):
            return {k: unsubst(v, tmp) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return [unsubst(v, tmp) for v in x]
        if isinstance(x, (str, Path)):
            s = str(x)
            for root in (str(tmp.resolve()), str(tmp)):
                if s.startswith(root):
                    return "@TMP" + s[len(root):]
            return s
        if hasattr(x, "kwargs") and type(x).__name__.endswith("Modifier"):
            return canon_recipe(x)
        return bc.jsonable(x)

    quant = {}
    for cid, m, model_id, kw in bc.QUANTIZE_CASES:
        tmp = Path(tempfile.mkdtemp(prefix="case_", dir=scratch))
        os.chdir(tmp)
        calls = []
        marker = object()
        fail = RuntimeError("engine exploded") if cid == "engine-fails" else None
        llmcompressor.CURRENT["oneshot"] = bc.recording_oneshot(calls, lambda: marker, fail)
        type(QuantizerRegistry.create(m, model_id="x"))._ONESHOT_PARAMS_CACHE = None
        from quantool.methods.llm_compressor.base import LLMCompressorQuantizer

        LLMCompressorQuantizer._ONESHOT_PARAMS_CACHE = None
        q = QuantizerRegistry.create(m, model_id=model_id)
        loguru.COUNTS.update(warning=0, error=0)
        res = {}
        try:
            ret = q.quantize(**subst(kw, tmp))
            res["returned"] = unsubst(ret, tmp)
        except Exception as e:  # noqa: BLE001
            res["raises"] = type(e).__name__
        res["errors_logged"] = loguru.COUNTS["error"]
        res["engine_calls"] = [unsubst(c, tmp) for c in calls]
        res["state"] = {
            "last_output_dir": unsubst(q.last_output_dir, tmp) if q.last_output_dir else None,
            "source_model": unsubst(q.source_model, tmp),
            "last_model_is_engine_result": q.last_model is marker,
            "last_recipe": unsubst(q._last_recipe, tmp) if q._last_recipe is not None else None,
        }
        res["dirs_created"] = sorted(str(p.relative_to(tmp)) for p in tmp.rglob("*") if p.is_dir())
        quant[cid] = res
    os.chdir(scratch)
    out["quantize"] = quant

    # ---- _default_output_dir ---------------------------------------------------------------------
    out["output_dirs"] = [
        {"method": m, "model_id": mid, "level_hint": lv,
         "dir": str(QuantizerRegistry.create(m, model_id=mid)._default_output_dir(lv))}
        for m, mid, lv in bc.OUTPUT_DIR_CASES]

    # ---- save hook without a model ---------------------------------------------------------------
    q = QuantizerRegistry.create("gptq", model_id="m")
    try:
        q._save_model_files(str(scratch / "nowhere"))
        out["save_without_model"] = None
    except Exception as e:  # noqa: BLE001
        out["save_without_model"] = type(e).__name__

    # ---- convert_row -----------------------------------------------------------------------------
    rows = {}
    for cid, row, tok, extra in bc.ROW_CASES:
        tokenizer = bc.TOKENIZERS[tok]()
        given = json.loads(json.dumps(row))
        try:
            got = dt.convert_row(given, tokenizer, **extra)
            rows[cid] = {"result": bc.jsonable(got), "same_object": got is given}
        except Exception as e:  # noqa: BLE001
            rows[cid] = {"raises": type(e).__name__}
    out["rows"] = rows
    helpers = {}
    for cid, row, tok, _ in bc.ROW_CASES:
        try:
            helpers[cid] = {"is_conversational": bool(dt.is_conversational(row))}
        except Exception as e:  # noqa: BLE001
            helpers[cid] = {"is_conversational_raises": type(e).__name__}
    out["row_helpers"] = helpers
    out["has_chat_template"] = {
        name: {"plain": dt.has_chat_template(cls()), "verify": dt.has_chat_template(cls(), verify=True)}
        for name, cls in bc.TOKENIZERS.items()}

    # ---- prepare_calibration_data ----------------------------------------------------------------
    prep = {}
    q = QuantizerRegistry.create("gptq", model_id="m")
    for cid, rws, tok in bc.PREPARE_CASES:
        ds = bc.Rows(rws)
        got = q.prepare_calibration_data(ds, tokenizer=bc.TOKENIZERS[tok]() if tok else None)
        prep[cid] = {"same_object": got is ds, "columns": list(got.column_names), "rows": bc.jsonable(got.rows)}
    out["prepare"] = prep

    # ---- provenance ------------------------------------------------------------------------------
    files = ["quantool/core/registry.py", "quantool/core/base.py", "quantool/methods/llm_compressor/base.py",
             "quantool/methods/llm_compressor/gptq/gptq.py", "quantool/methods/llm_compressor/awq/awq.py",
             "quantool/methods/llm_compressor/smoothquant/smoothquant.py", "quantool/utils/dataset_textifier.py"]
    out["provenance"] = {
        "generator": "tests/golden/make_boundary_fixtures.py",
        "reference_files_sha1": {f: hashlib.sha1((REF_SRC / f).read_bytes()).hexdigest() for f in files},
        "stand_ins": sorted(k.split("/")[0] for k in STUBS if k.count("/") == 1),
        "note": "outputs of the reference's own plumbing on this repo's probe inputs; the engine behind "
                "oneshot and the modifiers are recording stand-ins (not installed in the container)",
    }
    OUT.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes): {len(recipes)} recipe, {len(quant)} quantize, "
          f"{len(rows)} row, {len(prep)} prepare cases")


if __name__ == "__main__":
    main()
