"""engine/sequential.py:merge_cache -- host logic (CPU tensors): which cached layer inputs may share a forward."""
import torch

from quantool_amd.engine.sequential import merge_cache


def _entry(T, K=8, fill=0.0, pos_shift=0, flag=False):
    h = torch.full((1, T, K), fill)
    cos, sin = torch.ones(1, T, 4) * (1 + pos_shift), torch.zeros(1, T, 4)
    return ((h,), {"attention_mask": None, "position_ids": torch.arange(T).unsqueeze(0) + pos_shift,
                   "position_embeddings": (cos, sin), "cache_position": torch.arange(T), "use_cache": flag})


def test_equal_shapes_are_stacked_along_the_batch_dimension():
    cache = [_entry(6, fill=float(i)) for i in range(5)]
    out = merge_cache(cache, max_tokens=18)             # 3 samples of 6 tokens per forward
    assert [e[0][0].shape[0] for e in out] == [3, 2]
    (h,), kw = out[0]
    assert h.shape == (3, 6, 8) and [float(h[i, 0, 0]) for i in range(3)] == [0.0, 1.0, 2.0]   # order kept
    assert kw["position_ids"].shape == (3, 6) and kw["position_embeddings"][0].shape == (3, 6, 4)
    assert isinstance(kw["position_embeddings"], tuple) and kw["attention_mask"] is None
    assert kw["cache_position"].shape == (6,) and kw["use_cache"] is False      # no batch dimension: taken once


def test_different_shapes_or_settings_are_not_merged():
    cache = [_entry(6), _entry(6), _entry(4), _entry(4), _entry(6), _entry(6, flag=True)]
    out = merge_cache(cache, max_tokens=1000)
    assert [(e[0][0].shape[0], e[0][0].shape[1]) for e in out] == [(2, 6), (2, 4), (1, 6), (1, 6)]


def test_per_sample_values_travel_with_their_sample_and_shared_tensors_must_agree():
    a, b = _entry(5, fill=1.0, pos_shift=0), _entry(5, fill=2.0, pos_shift=3)
    out = merge_cache([a, b], max_tokens=1000)
    # position_ids / rotary tables have a batch dimension: each sample keeps its own
    assert len(out) == 1 and out[0][1]["position_ids"][1, 0] == 3 and float(out[0][1]["position_embeddings"][0][1, 0, 0]) == 4.0
    c = _entry(5)
    c[1]["cache_position"] = torch.arange(5) + 1        # a tensor WITHOUT a batch dimension that differs: no merge
    assert len(merge_cache([a, c], max_tokens=1000)) == 2


def test_zero_budget_keeps_one_sample_per_forward():
    cache = [_entry(6) for _ in range(3)]
    assert merge_cache(cache, 0) is cache
    assert len(merge_cache(cache, 6)) == 3              # budget of one sample
