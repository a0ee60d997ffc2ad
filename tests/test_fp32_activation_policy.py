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
