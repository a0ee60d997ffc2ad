#!/usr/bin/env python3
"""BASELINE.json's north-star job end to end: a random-init Llama-3-8B-SHAPED model (32 layers, hidden 4096,
intermediate 14336, 32 heads / 8 KV heads, vocab 128256; no download) quantised to GPTQ W4A16 g128 with
512 calibration samples x 384 random tokens through the quantool plugin API (`QuantizerRegistry.create("gptq")
.quantize(...)`), then saved as a compressed-tensors checkpoint and read back.

usage: full_model.py [layers [samples [seq [method [level [shape]]]]]]       (defaults: 32 512 384 gptq W4A16 8b)
shape mixtral = Mixtral-8x7B's dimensions (8 experts, top-2): e.g. `32 512 384 smoothquant W4A8 mixtral`.
shape 70b = Llama-3-70B's dimensions (hidden 8192, intermediate 28672, 64 heads / 8 KV heads; 80 layers = 141 GB of
bf16 weights, which one MI355X holds): the checkpoint then goes to QT_FULL_MODEL_OUT (default: a temp dir) once only.

Prints wall time of quantize(), of save_pretrained(), the checkpoint size, and a read-back check of one packed
Linear against the model's own written-back weight.  Numbers: profiles/r03_full_model.txt."""
import logging
import os
import sys
import tempfile
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("QT_CALIB_TIMING", "1")
import torch
from transformers import LlamaConfig, LlamaForCausalLM

import quantool_amd.methods  # noqa: F401
from quantool_amd.core import QuantizerRegistry

logging.basicConfig(level=logging.WARNING)
argv = sys.argv[1:]
layers = int(argv[0]) if len(argv) > 0 else 32
n_samples = int(argv[1]) if len(argv) > 1 else 512
seq = int(argv[2]) if len(argv) > 2 else 384
method = argv[3] if len(argv) > 3 else "gptq"
level = argv[4] if len(argv) > 4 else "W4A16"
shape = argv[5] if len(argv) > 5 else "8b"
dev = torch.device("cuda:0")

if shape == "mixtral":      # Mixtral-8x7B's dimensions: 8 experts, top-2 routing (BASELINE.json configs[4])
    from transformers import MixtralConfig, MixtralForCausalLM

    cfg = MixtralConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32,
                        num_key_value_heads=8, num_local_experts=8, num_experts_per_tok=2, vocab_size=32000,
                        max_position_embeddings=8192, rope_theta=1e6, rms_norm_eps=1e-5, tie_word_embeddings=False)
    model_cls = MixtralForCausalLM
else:
    hidden, inter, heads = {"8b": (4096, 14336, 32), "70b": (8192, 28672, 64)}[shape]
    cfg = LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                      num_key_value_heads=8, vocab_size=128256, max_position_embeddings=8192, rope_theta=500000.0,
                      rms_norm_eps=1e-5, tie_word_embeddings=False)
    model_cls = LlamaForCausalLM
t0 = time.perf_counter()
torch.manual_seed(0)
prev = torch.get_default_dtype()
torch.set_default_dtype(torch.bfloat16)
try:
    with torch.device(dev):
        model = model_cls(cfg)
finally:
    torch.set_default_dtype(prev)
model.eval()
torch.cuda.synchronize()
n_lin = sum(p.numel() for n_, p in model.named_parameters() if "proj" in n_ or "experts" in n_)
print(f"model: {layers} layers, {sum(p.numel() for p in model.parameters()) / 1e9:.2f} G parameters "
      f"({n_lin / 1e9:.2f} G in the decoder Linears), built in {time.perf_counter() - t0:.1f} s", flush=True)

g = torch.Generator().manual_seed(0)
data = [{"input_ids": torch.randint(0, cfg.vocab_size, (seq,), generator=g)} for _ in range(n_samples)]
probe = torch.randint(0, cfg.vocab_size, (1, 64), generator=g).to(dev)
with torch.no_grad():
    before = model(input_ids=probe).logits.float()

with tempfile.TemporaryDirectory(dir=os.environ.get("QT_FULL_MODEL_OUT")) as tmp:
    q = QuantizerRegistry.create(method, model_id=f"synthetic/{shape}-shaped")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    q.quantize(model=model, level=level, dataset=data, num_calibration_samples=n_samples, max_seq_length=seq,
               oneshot_kwargs={"output_dir": tmp + "/work"})
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{method} {level}: quantize() {dt:.2f} s wall = {n_lin / dt / 1e9:.3f} G weights/s "
          f"({dt / layers:.3f} s per decoder layer incl. the one-off input capture), "
          f"{n_samples} samples x {seq} tokens", flush=True)
    print(f"peak GPU memory {torch.cuda.max_memory_allocated(dev) / 2**30:.1f} GiB allocated, "
          f"{torch.cuda.max_memory_reserved(dev) / 2**30:.1f} GiB reserved", flush=True)
    saved = tmp + "/saved"
    if shape != "8b":           # one copy of a 25-40 GB checkpoint is enough: read back what quantize() wrote
        saved = tmp + "/work"
        t0 = time.perf_counter()
    else:
        t0 = time.perf_counter()
        q.save_pretrained(saved)
    ds = time.perf_counter() - t0
    files = sorted(Path(saved).glob("*"))
    size = sum(f.stat().st_size for f in files)
    print(f"save_pretrained(): {ds:.2f} s, {size / 1e9:.2f} GB in {len(files)} files "
          f"({', '.join(f.name for f in files[:6])}{' ...' if len(files) > 6 else ''})", flush=True)
    with torch.no_grad():
        after = model(input_ids=probe).logits.float()
    print(f"logits relative change on a 64-token probe: {float((after - before).norm() / before.norm()):.3f}", flush=True)
    if level.upper().startswith("W4") and shape != "mixtral":
        # read one packed Linear back and compare with what the driver wrote into the module
        import json

        from safetensors import safe_open

        name = f"model.layers.{layers - 1}.mlp.down_proj"
        idx = Path(saved) / "model.safetensors.index.json"
        fname = json.loads(idx.read_text())["weight_map"][name + ".weight_packed"] if idx.exists() else "model.safetensors"
        with safe_open(str(Path(saved) / fname), framework="pt") as f:      # only this Linear's tensors
            state = {k: f.get_tensor(k) for k in f.keys() if k.startswith(name + ".")}
        packed, scale = state[name + ".weight_packed"], state[name + ".weight_scale"].float()
        wshape = state[name + ".weight_shape"].tolist()
        nib = torch.stack([(packed >> (4 * j)) & 0xF for j in range(8)], dim=-1).reshape(packed.shape[0], -1)[:, :wshape[1]]
        w = (nib.to(torch.int32) - 8).float()
        gs = wshape[1] // scale.shape[1]
        w = (w.reshape(wshape[0], -1, gs) * scale[:, :, None]).reshape(wshape[0], wshape[1])
        if name + ".weight_g_idx" in state:
            print("(g_idx present: skipped the read-back compare)")
        else:
            ref = model.get_submodule(name).weight.detach().float().cpu()
            err = float((w - ref).abs().max())
            print(f"read-back of {name}: {tuple(wshape)}, max |dequant(packed) - written-back weight| = {err:.3e} "
                  f"(bf16 rounding of the write-back: <= {float(ref.abs().max()) * 2 ** -8:.1e})", flush=True)
