"""INPUTS of the boundary / front-end fixtures, shared by the generator (``make_boundary_fixtures.py``:
runs them through the REFERENCE's plumbing in the build container and stores what came out in
``boundary_reference.json``) and by the tests (``tests/test_boundary.py``, ``tests/test_calibration_text.py``:
run them through ``quantool_amd`` and compare).  Nothing here comes from the reference: these are
this repo's own probe inputs and stand-in objects.
"""
from __future__ import annotations

import inspect
from typing import Any, Dict, List

#: keyword names of the recording stand-in for the engine's ``oneshot`` -- BOTH sides route
#: ``quantize(**kwargs)`` by matching names against the engine function's signature
#: (reference ``base.py:45-72,117-124``), so both are given a function with exactly these names.
ONESHOT_PARAMS = ["model", "dataset", "recipe", "output_dir", "num_calibration_samples", "max_seq_length",
                  "shuffle_calibration_samples", "save_compressed", "trust_remote_code_model", "dataset_path",
                  "calibration_dataloader", "tokenizer", "splits", "text_column", "precision", "pipeline",
                  "sequential_targets"]

#: the preset names the reference lists as valid (``gptq.py:68-70`` / ``awq.py:62-64`` / ``smoothquant.py:66-68``)
PRESETS = ["W8A16", "W4A16", "W4A16_ASYM", "W8A8", "INT8", "W4A8", "FP8", "FP8_DYNAMIC", "FP8_BLOCK", "NVFP4A16",
           "NVFP4", "UNQUANTIZED"]


def recording_oneshot(calls: list, result_factory=lambda: object(), fail: Exception = None):
    """A function whose ``inspect.signature`` shows ONESHOT_PARAMS (all keyword, default None) and which
    appends the kwargs it was called with to ``calls``."""
    def oneshot(**kw):
        calls.append(dict(kw))
        if fail is not None:
            raise fail
        return result_factory()

    oneshot.__signature__ = inspect.Signature(
        [inspect.Parameter(n, inspect.Parameter.KEYWORD_ONLY, default=None) for n in ONESHOT_PARAMS])
    return oneshot


# ------------------------------------------------------------------------------------------------
# _build_recipe(level, method_kwargs) probes: (case id, method, level, method_kwargs)
# ------------------------------------------------------------------------------------------------
_ALL_KEYS = {"scheme": "W8A16", "targets": ["Linear", "re:.*proj"], "ignore": ["lm_head", "re:.*gate$"],
             "block_size": 64, "dampening_frac": 0.1, "sequential_targets": ["LlamaDecoderLayer"],
             "actorder": "group", "mappings": [["re:.*norm", ["re:.*q_proj"]]], "smoothing_strength": 0.7,
             "offload_hessians": True, "duo_scaling": False}

RECIPE_CASES: List[tuple] = []
for _m, _levels in (("gptq", ["W4A16", "W8A8", "INT8", "W8A16", "W4A16_ASYM", "W4A8"]),
                    ("awq", ["W4A16", "W4A16_ASYM", "W8A16"]),
                    ("smoothquant", ["W8A8", "INT8", "W4A8"])):
    RECIPE_CASES.append((f"{_m}-default", _m, None, {}))
    RECIPE_CASES.append((f"{_m}-empty-level", _m, "", {"scheme": "W8A16"}))
    for _l in _levels:
        RECIPE_CASES.append((f"{_m}-{_l}", _m, _l, {}))
    RECIPE_CASES.append((f"{_m}-scheme-from-kwargs", _m, None, {"scheme": "W8A16"}))
    RECIPE_CASES.append((f"{_m}-level-beats-kwargs", _m, "W4A16_ASYM", dict(_ALL_KEYS)))
    RECIPE_CASES.append((f"{_m}-all-keys", _m, None, dict(_ALL_KEYS)))
    RECIPE_CASES.append((f"{_m}-off-list-preset", _m, "FP8" if _m != "awq" else "W8A8", {}))
    RECIPE_CASES.append((f"{_m}-invalid-scheme", _m, "W3A16", {}))
    RECIPE_CASES.append((f"{_m}-invalid-scheme-kwargs", _m, None, {"scheme": "int4"}))
    RECIPE_CASES.append((f"{_m}-targets-string", _m, None, {"targets": "Linear", "ignore": []}))

# ------------------------------------------------------------------------------------------------
# quantize(...) probes: (case id, method, model_id, quantize kwargs).  "@DATASET" / "@RECIPE" / "@TMP/..."
# are replaced by a marker object / a path under the case's scratch directory on both sides.
# ------------------------------------------------------------------------------------------------
QUANTIZE_CASES: List[tuple] = [
    ("truth-table", "gptq", "org/model", dict(
        model="/local/model", level="W4A16", dataset="@DATASET", num_calibration_samples=32, max_seq_length=256,
        oneshot_kwargs={"max_seq_length": 128}, method_kwargs__block_size=64,
        targets="Embedding", ignore=["nothing"], bogus=1)),
    ("yaml-field-set", "gptq", "synthetic/opt-125m-shaped", dict(
        model="/models/local", level="W4A16", dataset="@DATASET", scheme="W8A8", targets="Linear", ignore=["lm_head"],
        num_calibration_samples=32)),
    ("dataset-path-only", "gptq", "m", dict(model="/local/model", level="W4A16", dataset_path="/data/calib.json")),
    ("dataset-path-in-oneshot-kwargs", "awq", "m", dict(model="x", oneshot_kwargs={"dataset_path": "/d.json"})),
    ("dataloader-in-oneshot-kwargs", "awq", "m", dict(model="x", oneshot_kwargs={"calibration_dataloader": "@DATASET"})),
    ("no-calibration", "gptq", "m", dict(model="/local/model", level="W4A16")),
    ("empty-dataset", "gptq", "m", dict(model="/local/model", level="W4A16", dataset=[])),
    ("empty-dataset-but-path", "gptq", "m", dict(model="/local/model", dataset=[], dataset_path="/d.json")),
    ("explicit-recipe", "gptq", "m", dict(model="x", recipe="@RECIPE", dataset="@DATASET")),
    ("explicit-recipe-with-level", "gptq", "m", dict(model="x", recipe="@RECIPE", level="W8A16", dataset="@DATASET")),
    ("recipe-in-oneshot-kwargs-wins", "gptq", "m", dict(model="x", level="W4A16", dataset="@DATASET",
                                                        oneshot_kwargs={"recipe": "from-oneshot-kwargs"})),
    ("output-dir-in-oneshot-kwargs", "gptq", "m", dict(model="x", level="W4A16", dataset="@DATASET",
                                                       oneshot_kwargs={"output_dir": "@TMP/custom/out"})),
    ("output-dir-as-keyword", "smoothquant", "m", dict(model="x", dataset="@DATASET", output_dir="@TMP/kw_out")),
    ("method-kwargs-dict-and-prefix", "gptq", "m", dict(
        model="x", dataset="@DATASET", method_kwargs={"scheme": "W8A16", "block_size": 32, "ignore": ["a"]},
        method_kwargs__block_size=256, method_kwargs__dampening_frac=0.05, method_kwargs__unknown="u")),
    ("awq-mappings-via-prefix", "awq", "org/m", dict(
        model="x", level="W4A16_ASYM", dataset="@DATASET", method_kwargs__mappings=[["a", ["b"]]],
        method_kwargs__smoothing_strength=0.3)),
    ("smoothquant-defaults", "smoothquant", "a/b/c", dict(model="x", dataset="@DATASET")),
    ("smoothquant-strength", "smoothquant", "m", dict(model="x", level="W4A8", dataset="@DATASET",
                                                      method_kwargs__smoothing_strength=0.8,
                                                      method_kwargs__block_size=64)),
    ("model-id-none", "gptq", None, dict(model="x", level="W4A16", dataset="@DATASET")),
    ("level-with-slash", "gptq", "m", dict(model="x", recipe="@RECIPE", level="a/b", dataset="@DATASET")),
    ("model-in-oneshot-kwargs-wins", "gptq", "m", dict(model="x", level="W4A16", dataset="@DATASET",
                                                       oneshot_kwargs={"model": "other", "save_compressed": False,
                                                                       "trust_remote_code_model": False})),
    ("keyword-does-not-override-oneshot-kwargs", "gptq", "m", dict(
        model="x", level="W4A16", dataset="@DATASET", precision="bf16", pipeline="sequential",
        oneshot_kwargs={"precision": "fp16"})),
    ("dataset-param-overrides-oneshot-kwargs", "gptq", "m", dict(
        model="x", level="W4A16", dataset="@DATASET", oneshot_kwargs={"dataset": "from-dict"})),
    ("invalid-scheme", "gptq", "m", dict(model="x", level="W3A16", dataset="@DATASET")),
    ("engine-fails", "gptq", "m", dict(model="x", level="W4A16", dataset="@DATASET")),
]


# ------------------------------------------------------------------------------------------------
# calibration front-end probes
# ------------------------------------------------------------------------------------------------
class MarkupTokenizer:
    """Deterministic chat-template stand-in: ``<s>role:content|role:content</s>`` with an open assistant
    turn for ``add_generation_prompt``, no closing marker for ``continue_final_message``; extra keyword
    arguments are rendered into the text so that their routing is visible in the output."""

    chat_template = "{{ messages }}"

    def apply_chat_template(self, messages, tools=None, tokenize=False, add_generation_prompt=False,
                            continue_final_message=False, **kw):
        if any("content" not in m for m in messages):
            raise ValueError("message without content")
        text = "<s>" + "|".join(f"{m['role']}:{m['content']}" for m in messages)
        if tools:
            text = f"[tools={len(tools)}]" + text
        if kw:
            text = "[" + ",".join(f"{k}={kw[k]}" for k in sorted(kw)) + "]" + text
        if continue_final_message:
            return text
        return text + ("</s>|assistant:" if add_generation_prompt else "</s>")


class NoTemplateTokenizer:
    chat_template = None

    def apply_chat_template(self, *a, **k):
        raise AssertionError("must not be called")


class PlainTokenizer:
    pass


def U(c):
    return {"role": "user", "content": c}


def A(c):
    return {"role": "assistant", "content": c}


def S(c):
    return {"role": "system", "content": c}


#: (case id, row, tokenizer name, extra convert_row kwargs)
ROW_CASES: List[tuple] = [
    ("messages", {"messages": [S("be brief"), U("hi"), A("hello")]}, "markup", {}),
    ("prompt-user", {"prompt": [U("What colour is the sky?")]}, "markup", {}),
    ("prompt-assistant-continue", {"prompt": [U("Count"), A("1, 2,")]}, "markup", {}),
    ("prompt-bad-last-role", {"prompt": [U("x"), S("y")]}, "markup", {}),
    ("prompt-completion", {"prompt": [U("2+2?")], "completion": [A("4")]}, "markup", {}),
    ("prompt-completion-label", {"prompt": [U("2+2?")], "completion": [A("5")], "label": False}, "markup", {}),
    ("preference", {"prompt": [U("pick")], "chosen": [A("good")], "rejected": [A("bad")]}, "markup", {}),
    ("implicit-preference", {"chosen": [U("q"), A("good")], "rejected": [U("q"), A("bad")]}, "markup", {}),
    ("template-kwargs-per-row", {"messages": [U("hi")], "chat_template_kwargs": {"flavour": "x"}}, "markup", {}),
    ("template-kwargs-call-beats-row", {"messages": [U("hi")], "chat_template_kwargs": {"flavour": "x"}}, "markup",
     {"flavour": "y", "mode": 2}),
    ("tools", {"prompt": [U("weather?")]}, "markup", {"tools": [{"name": "get_weather"}]}),
    ("extra-columns-dropped", {"messages": [U("hi")], "id": 7, "source": "s"}, "markup", {}),
    ("plain-text-row", {"text": "plain text"}, "markup", {}),
    ("io-row", {"input": "a", "output": "b"}, "markup", {}),
    ("string-prompt", {"prompt": "a string prompt", "completion": "a string"}, "markup", {}),
    ("no-template", {"messages": [U("hi")]}, "none", {}),
    ("plain-tokenizer", {"messages": [U("hi")]}, "plain", {}),
    ("invalid-messages-and-prompt", {"messages": [U("hi")], "prompt": [U("hi")]}, "markup", {}),
    ("invalid-completion-only", {"completion": [A("x")]}, "markup", {}),
    ("invalid-chosen-only", {"chosen": [A("x")]}, "markup", {}),
    ("invalid-prompt-chosen", {"prompt": [U("x")], "chosen": [A("y")]}, "markup", {}),
    ("invalid-label-with-preference", {"prompt": [U("x")], "chosen": [A("y")], "rejected": [A("z")], "label": True},
     "markup", {}),
    ("render-failure-returns-row", {"messages": [{"role": "user", "content": "ok"}, {"role": "assistant"}]}, "markup", {}),
]

TOKENIZERS = {"markup": MarkupTokenizer, "none": NoTemplateTokenizer, "plain": PlainTokenizer}


class Rows:
    """The two members of a HF ``Dataset`` the plugins touch: ``column_names`` and ``map``."""

    def __init__(self, rows: List[Dict[str, Any]]):
        self.rows = [dict(r) for r in rows]

    @property
    def column_names(self):
        cols: List[str] = []
        for r in self.rows:
            cols += [k for k in r if k not in cols]
        return cols

    def map(self, fn, batched=False):
        return Rows([{**r, **fn(r)} for r in self.rows])


#: (case id, rows, tokenizer name or None)
PREPARE_CASES: List[tuple] = [
    ("text-present", [{"text": "t", "prompt": "p"}], None),
    ("text-target-present", [{"text_target": "t"}], None),
    ("fallback-prompt", [{"prompt": "p", "completion": "c"}], None),
    ("fallback-completion", [{"completion": "c", "label": True}], None),
    ("fallback-chosen", [{"chosen": "c", "rejected": "r"}], None),
    ("fallback-rejected", [{"rejected": "r"}], None),
    ("fallback-label", [{"label": "l"}], None),
    ("nothing-usable", [{"other": 1}], None),
    ("chat-messages", [{"messages": [U("hi"), A("yo")]}], "markup"),
    ("chat-prompt-completion", [{"prompt": [U("2+2?")], "completion": [A("4")]}], "markup"),
    ("chat-preference", [{"prompt": [U("pick")], "chosen": [A("good")], "rejected": [A("bad")]}], "markup"),
    ("chat-plain-rows-with-tokenizer", [{"text": "x"}, {"text": "y"}], "markup"),
    ("chat-no-template", [{"prompt": [U("q")]}], "none"),
]

#: _default_output_dir probes: (method, model_id, level_hint)
OUTPUT_DIR_CASES = [("gptq", "org/model", "W4A16"), ("awq", "a/b/c", None), ("smoothquant", None, "W8A8"),
                    ("gptq", "", "x/y"), ("awq", "plain", "")]


def jsonable(x):
    """Canonical JSON form of a probe result (tuples -> lists, markers by name)."""
    if isinstance(x, dict):
        return {str(k): jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [jsonable(v) for v in x]
    if isinstance(x, (str, int, float, bool)) or x is None:
        return x
    return f"<{type(x).__name__}>"
