"""fp32 calibration activations (an fp32 checkpoint; the reference accumulates ``inp.float()``): the Gram
accumulation takes them through its fp32-accurate three-plane product by default (GPU test:
tests/test_gpu_fp32_activations.py); the statistics passes -- and the Gram pass under QT_FP32_ACTIVATIONS=bf16 --
round them to bf16, loudly and refusably.  Host logic only (the policy checks run before any device call)."""
import logging

import pytest
import torch

from quantool_amd.hip import ops


def test_16_bit_activations_pass_through_untouched():
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(4, 8).to(dt)
        assert ops.as_act16(x) is x


def test_fp32_activations_are_rounded_with_one_warning(caplog, monkeypatch):
    monkeypatch.delenv("QT_FP32_ACTIVATIONS", raising=False)
    monkeypatch.setattr(ops, "_FP32_ACT_WARNED", False)
    x = torch.randn(4, 8)
    with caplog.at_level(logging.WARNING):
        y = ops.as_act16(x)
        ops.as_act16(x)
    assert y.dtype == torch.bfloat16 and torch.equal(y, x.to(torch.bfloat16))
    msgs = [r for r in caplog.records if "rounded to bf16" in r.message]
    assert len(msgs) == 1                                   # once per process


def test_fp32_activations_can_be_refused(monkeypatch):
    monkeypatch.setenv("QT_FP32_ACTIVATIONS", "error")
    with pytest.raises(ValueError, match="16-bit activations"):
        ops.as_act16(torch.randn(4, 8))
    with pytest.raises(ValueError, match="16-bit activations"):
        ops.wide_activation_policy(torch.float32)
    assert ops.as_act16(torch.randn(4, 8).to(torch.bfloat16)).dtype == torch.bfloat16     # 16-bit inputs unaffected


def test_integer_activations_are_a_type_error():
    with pytest.raises(TypeError):
        ops.as_act16(torch.ones(4, 8, dtype=torch.int32))


def test_gram_mode_knob(monkeypatch):
    monkeypatch.delenv("QT_FP32_ACTIVATIONS", raising=False)
    assert ops.wide_gram_mode() == "exact"
    for v, want in (("bf16", "bf16"), ("warn", "bf16"), ("error", "error"), ("EXACT", "exact")):
        monkeypatch.setenv("QT_FP32_ACTIVATIONS", v)
        assert ops.wide_gram_mode() == want
    monkeypatch.setenv("QT_FP32_ACTIVATIONS", "fp8")
    with pytest.raises(ValueError):
        ops.wide_gram_mode()


def _patched_accumulator(monkeypatch, K=64, stage_tokens=4096):
    """A HessianAccumulator on the host with the two Gram entry points replaced by recorders (host logic only)."""
    from quantool_amd.engine import gptq_linear as gl

    calls = []
    monkeypatch.setattr(gl.ops, "xtx_accumulate", lambda rows, G: calls.append(("x16", rows.dtype, rows.shape[0])))
    monkeypatch.setattr(gl.ops, "xtx_accumulate_f32", lambda rows, G: calls.append(("f32", rows.dtype, rows.shape[0])))
    return gl.HessianAccumulator(K, "cpu", stage_tokens=stage_tokens), calls


def test_fp32_staging_is_sized_once(monkeypatch):
    """the fp32 path halved ``stage_tokens`` on every direct batch and after every
    ``release_stage()``; a few long batches walked it down to 128 rows, after which every short batch took the direct
    path (one launch and one read-modify-write of G per sample)."""
    monkeypatch.delenv("QT_FP32_ACTIVATIONS", raising=False)
    acc, calls = _patched_accumulator(monkeypatch)
    assert acc.stage_tokens == 4096
    for _ in range(5):                                       # long batches: direct launches
        acc.add(torch.randn(2048, 64))
        assert acc.stage_tokens == 2048                      # halved once: same bytes at 4 B / element
    assert calls == [("f32", torch.float32, 2048)] * 5
    for _ in range(3):                                       # short batches are staged, not launched
        acc.add(torch.randn(384, 64))
    assert len(calls) == 5 and acc._fill == 3 * 384
    acc.release_stage()
    assert calls[-1] == ("f32", torch.float32, 3 * 384)
    acc.add(torch.randn(384, 64))                            # the first add after release_stage() does not halve again
    assert acc.stage_tokens == 2048 and acc._stage.shape[0] == 2048 and len(calls) == 6
    assert acc.n == 5 + 3 + 1


def test_wide_batch_after_16_bit_batches_goes_through_the_policy(monkeypatch, caplog):
    """Once the accumulator's dtype is bf16, a later fp32 batch is rounded to it -- logged once, or refused; never
    silently, and never by flipping a half-filled 16-bit accumulator to fp32."""
    monkeypatch.delenv("QT_FP32_ACTIVATIONS", raising=False)
    monkeypatch.setattr(ops, "_FP32_ACT_WARNED", False)
    acc, calls = _patched_accumulator(monkeypatch)
    acc.add(torch.randn(384, 64).to(torch.bfloat16))
    assert acc.dtype == torch.bfloat16
    with caplog.at_level(logging.WARNING):
        acc.add(torch.randn(384, 64))
    assert acc.dtype == torch.bfloat16 and acc.stage_tokens == 4096 and acc._stage.dtype == torch.bfloat16
    assert len([r for r in caplog.records if "rounded to bf16" in r.message]) == 1
    monkeypatch.setenv("QT_FP32_ACTIVATIONS", "error")
    with pytest.raises(ValueError, match="16-bit activations"):
        acc.add(torch.randn(384, 64))
    acc.flush()
    assert calls == [("x16", torch.bfloat16, 2 * 384)]


def test_launch_token_limit_splits_a_gram_pass_into_two_level_sums(monkeypatch):
    """QT_XTX_LAUNCH_TOKENS=n: at most n tokens per Gram launch (a two-level fp32 sum); 0 / unset: one
    launch per staged buffer or direct batch."""
    acc, calls = _patched_accumulator(monkeypatch, stage_tokens=1024)
    acc.add(torch.randn(2000, 64).to(torch.bfloat16))                   # direct batch, no limit: one launch
    assert calls == [("x16", torch.bfloat16, 2000)]
    monkeypatch.setenv("QT_XTX_LAUNCH_TOKENS", "700")                   # rounded down to a multiple of 64: 640
    acc.add(torch.randn(2000, 64).to(torch.bfloat16))
    assert [c[2] for c in calls[1:]] == [640, 640, 640, 80]
    for _ in range(3):                                                  # staged batches flush through the same limit
        acc.add(torch.randn(300, 64).to(torch.bfloat16))
    acc.flush()
    assert [c[2] for c in calls[5:]] == [640, 260]
    monkeypatch.setenv("QT_XTX_LAUNCH_TOKENS", "not-a-number")
    acc.add(torch.randn(2000, 64).to(torch.bfloat16))
    assert calls[-1][2] == 2000
