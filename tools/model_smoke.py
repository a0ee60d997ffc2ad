#!/usr/bin/env python3
"""Mid-size end-to-end run of the three plugins on a random-init Llama (no download): a crash / timing
check of the nn.Module path at shapes larger than the unit tests use."""
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from transformers import LlamaConfig, LlamaForCausalLM

import quantool_amd.methods  # noqa: F401
from quantool_amd.core import QuantizerRegistry

dev = torch.device("cuda:0")
hidden, inter, layers = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1024, 2816, 2)))
n_samples, seq = (int(x) for x in (sys.argv[4:6] if len(sys.argv) > 5 else (16, 128)))   # e.g. 512 384: the bench's calibration set
g = torch.Generator().manual_seed(0)
data = [{"input_ids": torch.randint(0, 1000, (seq,), generator=g)} for _ in range(n_samples)]
probe = torch.randint(0, 1000, (1, 64), generator=g).to(dev)
only = sys.argv[6].split(",") if len(sys.argv) > 6 else None      # e.g. "gptq" or "gptq,smoothquant"
import logging
logging.basicConfig(level=logging.WARNING)
for method, level in (("gptq", "W4A16"), ("awq", "W4A16"), ("smoothquant", "W8A8"), ("awq", "W8A16")):
    if only and method not in only:
        continue
    cfg = LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=8,
                      num_key_value_heads=8, vocab_size=1000, max_position_embeddings=max(256, seq), tie_word_embeddings=False)
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).to(torch.bfloat16).to(dev)
    with torch.no_grad():
        before = model(input_ids=probe).logits.float()
    with tempfile.TemporaryDirectory() as tmp:
        q = QuantizerRegistry.create(method, model_id="synthetic/mid-llama")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        q.quantize(model=model, level=level, dataset=data, num_calibration_samples=n_samples, max_seq_length=seq,
                   oneshot_kwargs={"output_dir": tmp})
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        q.save_pretrained(tmp + "/saved")
    with torch.no_grad():
        after = model(input_ids=probe).logits.float()
    rel = float((after - before).norm() / before.norm())
    n = sum(p.numel() for n_, p in model.named_parameters() if "proj" in n_)
    print(f"{method:12s} {level:6s}: {dt:6.2f} s, {n / dt / 1e6:8.1f} M weights/s, logits rel. change {rel:.3f}", flush=True)
