"""N3: calibration rows -> text.  The cases are the ones quantool's own suite pins for its
``convert_row`` (``tests/quantool/utils/test_dataset_textifier.py``: six row shapes, the pass-through
rules, the invalid key combinations), restated against this repo's converter, plus the split-at-common-
prefix property and the plugin hook that uses it."""
import pytest

from quantool_amd.utils import calibration_text as ct


class TemplateTok:
    """Renders `role:content` turns joined by `|` between fixed markers; records the flags it saw."""

    chat_template = "{{ messages }}"

    def __init__(self):
        self.calls = []

    def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=False, **kw):
        self.calls.append(dict(n=len(messages), add_generation_prompt=add_generation_prompt, **kw))
        text = "<s>" + "|".join(f"{m.get('role', 'user')}:{m.get('content', '')}" for m in messages)
        if kw.get("continue_final_message"):
            return text
        return text + ("</s>|assistant:" if add_generation_prompt else "</s>")


class PlainTok:
    pass


class BlankTemplateTok(TemplateTok):
    chat_template = "   "


U = lambda c: {"role": "user", "content": c}        # noqa: E731
A = lambda c: {"role": "assistant", "content": c}   # noqa: E731


def test_has_chat_template():
    assert ct.has_chat_template(TemplateTok()) is True
    assert ct.has_chat_template(TemplateTok(), verify=True) is True
    assert ct.has_chat_template(PlainTok()) is False
    assert ct.has_chat_template(PlainTok(), verify=True) is False
    assert ct.has_chat_template(BlankTemplateTok()) is False          # blank string: not a template ...
    assert ct.has_chat_template(BlankTemplateTok(), verify=True) is True   # ... unless a trial render works


@pytest.mark.parametrize("row,expected", [
    ({"messages": [U("hi")]}, True),
    ({"prompt": [U("hi")]}, True),
    ({"chosen": [A("yes")]}, True),
    ({"rejected": [A("no")]}, True),
    ({"completion": [A("done")]}, True),
    ({"text": "plain text"}, False),
    ({"input": "some input", "output": "some output"}, False),
    ({"prompt": "a string prompt"}, False),
    ({"messages": []}, False),
])
def test_is_conversational(row, expected):
    assert ct.is_conversational(row) is expected


def test_messages_row_becomes_text():
    out = ct.convert_row({"messages": [U("hi"), A("hello")]}, TemplateTok())
    assert out == {"text": "<s>user:hi|assistant:hello</s>"}


def test_prompt_completion_split_is_exact():
    tok = TemplateTok()
    row = {"prompt": [U("q?")], "completion": [A("a.")]}
    out = ct.convert_row(row, tok)
    assert set(out) == {"prompt", "completion"}
    full = TemplateTok().apply_chat_template(row["prompt"] + row["completion"])
    assert out["prompt"] + out["completion"] == full                 # nothing lost or duplicated at the seam
    assert out["prompt"] == "<s>user:q?" and out["completion"] == "|assistant:a.</s>"
    # the prompt alone was rendered with a generation prompt because it ends on a user turn
    assert tok.calls[0]["add_generation_prompt"] is True and tok.calls[0]["continue_final_message"] is False


def test_prompt_ending_on_assistant_continues_the_turn():
    tok = TemplateTok()
    out = ct.convert_row({"prompt": [U("q?"), A("The answer is")]}, tok)
    assert out == {"prompt": "<s>user:q?|assistant:The answer is"}
    assert tok.calls[0]["continue_final_message"] is True and tok.calls[0]["add_generation_prompt"] is False


def test_preference_rows():
    out = ct.convert_row({"prompt": [U("Which?")], "chosen": [A("ChoiceA")], "rejected": [A("ChoiceB")]}, TemplateTok())
    assert set(out) == {"prompt", "chosen", "rejected"}
    assert out["chosen"] != out["rejected"]
    assert out["prompt"] + out["chosen"] == "<s>user:Which?|assistant:ChoiceA</s>"
    out = ct.convert_row({"chosen": [A("Yes")], "rejected": [A("No")]}, TemplateTok())      # implicit prompt
    assert out == {"chosen": "<s>assistant:Yes</s>", "rejected": "<s>assistant:No</s>"}


def test_label_passes_through():
    out = ct.convert_row({"prompt": [U("q?")], "completion": [A("a.")], "label": "positive"}, TemplateTok())
    assert set(out) == {"prompt", "completion", "label"} and out["label"] == "positive"


def test_rows_returned_untouched():
    row = {"messages": [U("ping")]}
    assert ct.convert_row(row, PlainTok()) is row                     # tokenizer cannot render
    row = {"text": "plain text", "label": 1}
    assert ct.convert_row(row, TemplateTok()) is row                  # nothing conversational in it
    row = {"prompt": [{"role": "system", "content": "s"}]}            # template refuses: last turn is neither
    assert ct.convert_row(row, TemplateTok()) is row                  # user nor assistant -> row as is

    class Broken(TemplateTok):
        def apply_chat_template(self, *a, **k):
            raise RuntimeError("template error")

    row = {"messages": [U("x")]}
    assert ct.convert_row(row, Broken()) is row


@pytest.mark.parametrize("row", [
    {"prompt": [U("q?")], "messages": [U("hi")]},
    {"chosen": [A("yes")]},                      # rejected missing
    {"completion": [A("done")]},                 # prompt missing
    {"prompt": [U("q")], "label": True},         # label only goes with prompt + completion
])
def test_invalid_key_combinations_raise_key_error(row):
    with pytest.raises(KeyError):
        ct.convert_row(row, TemplateTok())


def test_template_kwargs_merge_call_site_wins():
    tok = TemplateTok()
    ct.convert_row({"messages": [U("x")], "chat_template_kwargs": {"enable_thinking": True, "date": "row"}}, tok,
                   tools=[{"name": "t"}], date="call")
    call = tok.calls[0]
    assert call["enable_thinking"] is True and call["date"] == "call" and call["tools"] == [{"name": "t"}]


def test_plugin_prepare_calibration_data_renders_conversations():
    import quantool_amd.methods  # noqa: F401
    from quantool_amd.core import QuantizerRegistry

    class DS:
        def __init__(self, rows):
            self.rows = rows
            self.column_names = sorted({k for r in rows for k in r})

        def map(self, fn, batched=False):
            return DS([{**r, **fn(r)} for r in self.rows])

    q = QuantizerRegistry.create("gptq", model_id="m")
    ds = q.prepare_calibration_data(DS([{"messages": [U("hi"), A("hello")]}]), tokenizer=TemplateTok())
    assert ds.rows[0]["text"] == "<s>user:hi|assistant:hello</s>"
    # prompt/completion rows have no `text` after templating: the first fallback column supplies it
    ds = q.prepare_calibration_data(DS([{"prompt": [U("q?")], "completion": [A("a.")]}]), tokenizer=TemplateTok())
    assert ds.rows[0]["text"] == "<s>user:q?" and ds.rows[0]["completion"] == "|assistant:a.</s>"
