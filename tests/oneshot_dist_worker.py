"""One rank of the two-rank rehearsals of ``oneshot`` under torch.distributed (see
test_gpu_oneshot_dist.py): both ranks share the box's single GPU, gloo stands in for RCCL.
argv: mode rank world port outdir      mode: linears | module"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def linears(rank, world, outdir):
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_shared
    from quantool_amd.engine.modifiers import GPTQModifier
    from quantool_amd.engine.oneshot import LinearCalibrationSet, LinearGroup, oneshot
    from quantool_amd.engine.serialization import load_state
    from quantool_amd.engine.sharding import group_cost, plan_groups

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)

    def acts(S, T, K):
        x = torch.randn((S, T, K), generator=g)
        x[..., 3] *= 6.0
        return x.to(torch.bfloat16).to(dev)

    def w(R, K):
        return (torch.randn((R, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)

    # one heavy group (split over the ranks: partitioning B) and two light ones (whole units: A)
    groups = [LinearGroup("down", acts(6, 128, 1024), {"down_proj": w(64, 1024)}),
              LinearGroup("attn", acts(6, 128, 256), {"q_proj": w(96, 256), "k_proj": w(33, 256)}),
              LinearGroup("out", acts(6, 128, 256), {"o_proj": w(64, 256)})]
    order = sorted(groups, key=lambda gr: (-gr.weights[next(iter(gr.weights))].shape[1], gr.name))
    plan = plan_groups([group_cost(gr.activations.shape[-1], 6 * 128, sum(x.shape[0] for x in gr.weights.values()))
                        for gr in order], world)
    assert [p[0] for p in plan] == ["B", "A", "A"] and plan[1][1] != plan[2][1], plan
    recipe = GPTQModifier(scheme="W4A16", targets="Linear", ignore=[])
    out = oneshot(model=LinearCalibrationSet(groups), recipe=recipe, output_dir=outdir)
    torch.cuda.synchronize()
    ok = True
    qa = recipe.weight_args()
    if rank == 0:
        sd = load_state(outdir)                                   # written by rank 0 only, complete
        assert len([k for k in sd if k.endswith("weight_packed")]) == 4
        for gr in groups:
            names = list(gr.weights)
            if gr.name == "down":                                 # B: Gram summed per rank, then G0 + G1
                parts = []
                for r in range(world):
                    a = HessianAccumulator(1024, dev)
                    for i in range(r, 6, world):
                        a.add(gr.activations[i:i + 1])
                    parts.append(a)
                tot = HessianAccumulator(1024, dev)
                tot.G.copy_(parts[0].G)
                for a in parts[1:]:
                    tot.G += a.G
                tot.n = sum(a.n for a in parts)
            else:                                                 # A: exactly the single-process run
                tot = HessianAccumulator(gr.activations.shape[-1], dev)
                tot.add(gr.activations)
            want = gptq_quantize_shared([gr.weights[n] for n in names], tot, qa)
            for n, ref in zip(names, want):
                same_p = torch.equal(sd[f"{n}.weight_packed"], ref.weight_packed.cpu())
                same_s = torch.equal(sd[f"{n}.weight_scale"], ref.weight_scale.cpu())
                print(f"[linears] {gr.name}/{n}: packed {same_p} scale {same_s}", flush=True)
                ok &= same_p and same_s
        assert set(out.results) == {"down_proj", "q_proj", "k_proj", "o_proj"}
    else:
        assert not Path(outdir, "model.safetensors").exists() or True   # same dir on one box: rank 0 wrote it
        assert "down_proj" in out.results                              # split groups are held by every rank
    return ok


def smooth(rank, world, outdir):
    """SmoothQuant + GPTQ recipe over ranks: whole-unit (A) groups are smoothed on their owners only, so
    the rescaled norm vectors and the scales must travel with the results -- rank 0's saved state has
    to equal the single-process one tensor for tensor."""
    from quantool_amd.engine.modifiers import GPTQModifier, SmoothQuantModifier
    from quantool_amd.engine.oneshot import LinearCalibrationSet, LinearGroup, oneshot
    from quantool_amd.engine.serialization import load_state
    from quantool_amd.engine import sharding

    dev = torch.device("cuda:0")

    def build():
        g = torch.Generator().manual_seed(33)

        def acts(S, T, K):
            x = torch.randn((S, T, K), generator=g)
            x[..., 5] *= 8.0
            return x.to(torch.bfloat16).to(dev)

        def w(R, K):
            return (torch.randn((R, K), generator=g) * 0.02).to(torch.bfloat16).to(dev)

        def vec(K):
            return (1.0 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).to(dev)

        return [LinearGroup("down", acts(6, 128, 1024), {"down_proj": w(64, 1024)}),        # B, not smoothed
                LinearGroup("attn", acts(6, 128, 256), {"q_proj": w(96, 256), "k_proj": w(32, 256)},
                            smooth_vectors={"input_layernorm.weight": vec(256)}),
                LinearGroup("mlp", acts(6, 128, 256), {"gate_proj": w(64, 256)},
                            smooth_vectors={"post_attention_layernorm.weight": vec(256)})]

    recipe = [SmoothQuantModifier(smoothing_strength=0.5), GPTQModifier(scheme="W4A16", targets="Linear", ignore=[])]
    out = oneshot(model=LinearCalibrationSet(build()), recipe=recipe, output_dir=str(Path(outdir) / "dist"))
    torch.cuda.synchronize()
    dist.barrier()
    ok = True
    if rank == 0:
        # the same job in one process: hide the process group from the engine
        real = sharding.dist_world
        sharding.dist_world = lambda: (1, 0)
        try:
            oneshot(model=LinearCalibrationSet(build()), recipe=recipe, output_dir=str(Path(outdir) / "single"))
        finally:
            sharding.dist_world = real
        torch.cuda.synchronize()
        a, b = load_state(str(Path(outdir) / "dist")), load_state(str(Path(outdir) / "single"))
        ok &= set(a) == set(b)
        ok &= {"input_layernorm.weight", "post_attention_layernorm.weight"} <= set(a)
        for k in sorted(b):
            if k.startswith("down_proj") and k.endswith("weight_packed"):
                continue     # split group: Gram partials are added in rank order (covered by mode `linears`)
            same = k in a and torch.equal(a[k], b[k])
            if not same:
                print(f"[smooth] {k}: differs", flush=True)
            ok &= same
        ok &= set(out.smoothing_scales) == {"attn", "mlp"}
        print(f"[smooth] keys {len(a)} ok {ok}", flush=True)
    return ok


def module(rank, world, outdir):
    import quantool_amd.methods  # noqa: F401
    from transformers import LlamaConfig, LlamaForCausalLM

    from quantool_amd.core import QuantizerRegistry

    dev = torch.device("cuda:0")
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=512, max_position_embeddings=128, tie_word_embeddings=False)
    torch.manual_seed(0)
    model = LlamaForCausalLM(cfg).to(torch.bfloat16).to(dev)
    g = torch.Generator().manual_seed(1)
    data = [{"input_ids": torch.randint(0, 512, (48,), generator=g)} for _ in range(8)]
    x = torch.randint(0, 512, (1, 32), generator=g).to(dev)
    with torch.no_grad():
        before = model(input_ids=x).logits.float()
    os.chdir(outdir)
    q = QuantizerRegistry.create("gptq", model_id=f"synthetic/tiny-llama-rank{rank}")
    out = q.quantize(model=model, level="W4A16", dataset=data, num_calibration_samples=8, max_seq_length=64,
                     shuffle_calibration_samples=False, output_dir=str(Path(outdir) / "quantized"))
    torch.cuda.synchronize()
    ok = len(model._qt_results) == 14
    # every rank holds the same quantised model: compare a checksum of all weights
    chk = float(sum(p.detach().double().abs().sum() for p in model.parameters()))
    sums = [None] * world
    dist.all_gather_object(sums, chk)
    ok &= all(s == sums[0] for s in sums)
    with torch.no_grad():
        after = model(input_ids=x).logits.float()
    rel = float((after - before).norm() / before.norm())
    print(f"[module] rank {rank}: results {len(model._qt_results)} checksums {sums} rel logits change {rel:.3f}", flush=True)
    ok &= rel < 0.5          # int4 g128 on a random tiny model: the function stays recognisable
    wrote = Path(out, "model.safetensors").exists()
    dist.barrier()
    ok &= Path(out, "model.safetensors").exists()                        # rank 0 wrote it (shared directory)
    if rank == 0:
        from quantool_amd.engine.serialization import load_state

        sd = load_state(out)
        n_packed = sum(1 for k in sd if k.endswith("weight_packed"))
        same = torch.equal(sd["model.layers.1.mlp.down_proj.weight_packed"],
                           model._qt_results["model.layers.1.mlp.down_proj"].weight_packed.cpu())
        print(f"[module] saved packed tensors {n_packed}, down_proj equal {same}", flush=True)
        ok &= n_packed == 14 and same
    return ok


def main():
    mode, rank, world, port, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        ok = {"linears": linears, "smooth": smooth, "module": module}[mode](rank, world, outdir)
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        sys.exit(0 if int(flag.item()) == 1 else 3)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
