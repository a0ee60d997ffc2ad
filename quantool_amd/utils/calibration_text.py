"""Calibration rows -> text.

``prepare_calibration_data`` (``src/quantool/methods/llm_compressor/base.py:257-345``) renders
conversational dataset rows through the tokenizer's chat template before ``oneshot`` tokenises the
``text`` column; quantool does that with ``quantool.utils.dataset_textifier.convert_row``
(``dataset_textifier.py:178-260``, itself modelled on TRL's ``apply_chat_template``).  This module
gives the MI355X backend the same row conversion without importing quantool, specified by the
behaviour quantool's own tests pin (``tests/quantool/utils/test_dataset_textifier.py``):

===============================  ==========================================================
row keys                         result keys
===============================  ==========================================================
``messages``                     ``text``
``prompt``                       ``prompt``
``prompt, completion[, label]``  ``prompt, completion[, label]``
``prompt, chosen, rejected``     ``prompt, chosen, rejected``
``chosen, rejected``             ``chosen, rejected`` (each rendered on its own)
anything else conversational     ``KeyError``
not conversational / no          the row object itself, untouched
chat template on the tokenizer
===============================  ==========================================================

A row that has a prompt is rendered twice per response -- prompt alone, prompt + response -- and split
at the longest common prefix, so that ``prompt + response`` is exactly the full rendering.  Any
failure inside the tokenizer's template call returns the row untouched.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, FrozenSet, List, Optional

_ROLE_KEYS = ("prompt", "chosen", "rejected", "completion", "messages")
_ALLOWED: List[FrozenSet[str]] = [frozenset(s) for s in (
    {"messages"}, {"prompt"}, {"prompt", "completion"}, {"prompt", "chosen", "rejected"},
    {"chosen", "rejected"}, {"prompt", "completion", "label"})]
_RESPONSE_ORDER = ("chosen", "rejected", "completion")


def has_chat_template(tokenizer: Any, verify: bool = False) -> bool:
    """True when ``tokenizer`` (or processor) can render conversations: it has ``apply_chat_template``
    and a non-blank ``chat_template`` string -- or, with ``verify``, a trial rendering succeeds."""
    render = getattr(tokenizer, "apply_chat_template", None)
    if render is None:
        return False
    template = getattr(tokenizer, "chat_template", None)
    if isinstance(template, str) and template.strip():
        return True
    if not verify:
        return False
    try:
        render([{"role": "user", "content": "ping"}], tokenize=False, add_generation_prompt=False)
    except Exception:  # noqa: BLE001
        return False
    return True


def is_conversational(row: Any) -> bool:
    """A row is conversational when one of its prompt / chosen / rejected / completion / messages
    fields is a list whose first element looks like ``{"role": ..., "content": ...}``.  Only one of
    the present fields is inspected, as quantool does."""
    present = [k for k in _ROLE_KEYS if k in row.keys()]
    if not present:
        return False
    value = row[present[-1]]
    if not isinstance(value, list) or not value:
        return False
    head = value[0]
    return isinstance(head, dict) and "role" in head and "content" in head


def _shared_prefix(a: str, b: str) -> str:
    n = 0
    for ca, cb in zip(a, b):
        if ca != cb:
            break
        n += 1
    return a[:n]


class _Renderer:
    def __init__(self, tokenizer, tools, kwargs):
        self._apply = tokenizer.apply_chat_template
        self._tools = tools
        self._kwargs = kwargs

    def __call__(self, messages, **flags) -> str:
        return self._apply(messages, tools=self._tools, tokenize=False, **flags, **self._kwargs)

    def prompt(self, messages) -> str:
        role = messages[-1].get("role")
        if role == "user":
            return self(messages, continue_final_message=False, add_generation_prompt=True)
        if role == "assistant":
            return self(messages, continue_final_message=True, add_generation_prompt=False)
        raise ValueError(f"a prompt must end with a user or assistant turn, not {role!r}")


def _convert(row: dict, render: _Renderer) -> Dict[str, Any]:
    if "messages" in row:
        return {"text": render(row["messages"], add_generation_prompt=False)}
    out: Dict[str, Any] = {}
    if "prompt" in row:
        prompt = render.prompt(row["prompt"])
        for key in _RESPONSE_ORDER:
            if key not in row:
                continue
            full = render(row["prompt"] + row[key])
            prompt = _shared_prefix(prompt, full)
            out[key] = full[len(prompt):]
        out["prompt"] = prompt
    else:
        for key in ("chosen", "rejected"):
            if key in row:
                out[key] = render(row[key])
    if "label" in row:
        out["label"] = row["label"]
    return out


def convert_row(example: dict, tokenizer: Any, tools: Optional[List[Any]] = None, **template_kwargs) -> dict:
    """Render one dataset row as the table in the module docstring says."""
    if not is_conversational(example) or not has_chat_template(tokenizer):
        return example
    keys = frozenset(k for k in example.keys() if k in _ROLE_KEYS or k == "label")
    if keys not in _ALLOWED:
        raise KeyError(f"unsupported combination of conversational fields: {sorted(keys)}")
    per_row = example.get("chat_template_kwargs") or {}
    try:
        return _convert(example, _Renderer(tokenizer, tools, {**per_row, **template_kwargs}))
    except Exception:  # noqa: BLE001 - a template that cannot render leaves the row for the text fallback
        return example


def row_converter() -> Callable[..., dict]:
    """The row converter the plugins use: always this module's (its outputs are pinned against quantool's own
    ``convert_row`` by the reference-generated fixtures, ``tests/golden/boundary_reference.json``), so the front-end
    does not change with what happens to be installed next to it."""
    return convert_row
