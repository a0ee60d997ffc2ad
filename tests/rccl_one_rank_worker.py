"""The collectives of SURVEY 8e over RCCL itself, with the one rank a one-GPU box allows (see
test_gpu_rccl_one_rank.py).  A one-rank group moves no bytes between GPUs, but every call goes through the
``nccl`` backend: tensors must be on the device, dtypes must be ones RCCL reduces / gathers, and the code around
each collective (buffer layouts, header flags, the table of contents of the final gather) runs as it will with
eight ranks.  The two-rank rehearsals (gloo, host tensors) cannot show any of that.   argv: port outdir"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    port = int(sys.argv[1])
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_shared
    from quantool_amd.engine.schemes import QuantArgs
    from quantool_amd.engine.serialization import result_tensors
    from quantool_amd.engine.sharding import (allreduce_accumulator, allreduce_gram, allreduce_inplace, gather_state_dict,
                                              gptq_quantize_row_split, gptq_quantize_token_split)

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    ok = True
    try:
        assert dist.get_backend() == "nccl"
        g = torch.Generator().manual_seed(11)
        K = 384
        X = torch.randn((6, 100, K), generator=g).to(torch.bfloat16).to(dev)
        X[..., 5] *= 8.0
        Ws = [(torch.randn((r, K), generator=g) * 0.02).to(torch.bfloat16).to(dev) for r in (64, 37, 1)]

        # -- the exchange step: Gram bands + sample count in one fp32 all-reduce
        acc = HessianAccumulator(K, dev)
        acc.add(X)
        before = acc.G.clone()
        n = allreduce_gram(acc.G, acc.n)
        same = torch.equal(torch.tril(acc.G), torch.tril(before))
        print(f"[rccl] allreduce_gram: lower triangle unchanged {same}, count {n}", flush=True)
        ok &= same and n == 6
        acc2 = HessianAccumulator(K, dev)
        acc2.add(X)
        allreduce_accumulator(acc2)
        ok &= acc2.n == 6 and torch.equal(torch.tril(acc2.G), torch.tril(before))
        t = torch.tensor([3.5, -1.0], dtype=torch.float64, device=dev)
        allreduce_inplace(t, op=dist.ReduceOp.MAX)
        ok &= t.tolist() == [3.5, -1.0]
        flag = torch.tensor([7], dtype=torch.int64, device=dev)
        allreduce_inplace(flag, op=dist.ReduceOp.MIN)
        ok &= int(flag.item()) == 7

        # -- partitioning B: row-split sweep + ONE uint8 all-gather, against the unsplit run
        for actorder, symmetric in (("static", True), ("group", False), (None, True)):
            qa = QuantArgs(num_bits=4, symmetric=symmetric, group_size=128, actorder=actorder)
            got = gptq_quantize_token_split(Ws, [X[i:i + 1] for i in range(6)], qa)
            ref_acc = HessianAccumulator(K, dev)
            for i in range(6):
                ref_acc.add(X[i:i + 1])
            want = gptq_quantize_shared(Ws, ref_acc, qa)
            for parts, ref in zip(got, want):
                ok &= torch.equal(parts["weight_packed"], ref.weight_packed)
                ok &= torch.equal(parts["weight_scale"], ref.weight_scale)
                if not symmetric:
                    ok &= torch.equal(parts["weight_zero_point"], ref.weight_zero_point)
                if actorder == "group":
                    ok &= torch.equal(parts["weight_g_idx"], ref.weight_g_idx)
            print(f"[rccl] token split actorder={actorder} symmetric={symmetric}: ok so far {ok}", flush=True)
        qa8 = QuantArgs(num_bits=8, symmetric=True, group_size=None, strategy="channel", actorder=None)
        acc8 = HessianAccumulator(K, dev)
        acc8.add(X)
        got8 = gptq_quantize_row_split(Ws[:2], acc8, qa8, with_dequantized=True)
        want8 = gptq_quantize_shared(Ws[:2], acc8, qa8)
        for a, b in zip(got8, want8):
            ok &= torch.equal(a.weight_q, b.weight_q) and torch.equal(a.weight_scale, b.weight_scale)
            ok &= a.dequantized().shape == b.weight_q.shape
        print(f"[rccl] row split W8 channel-wise with dequantised rows: ok so far {ok}", flush=True)

        # -- the final gather: int64[2] size exchange + payload with its table of contents
        qa = QuantArgs(num_bits=4, symmetric=True, group_size=128, actorder="static")
        res = gptq_quantize_shared(Ws[:1], acc, qa)[0]
        state = {f"layers.0.q_proj.{k}": v for k, v in result_tensors(res).items()}     # device + host tensors
        merged = gather_state_dict(state, dst=0, device=dev)
        ok &= set(merged) == set(state)
        ok &= all(torch.equal(merged[k].cpu(), state[k].cpu()) for k in state)
        ok &= all(v.device.type == "cuda" for v in merged.values())
        empty = gather_state_dict({}, dst=0, device=dev)
        ok &= empty == {}
        print(f"[rccl] gather_state_dict: {len(merged)} tensors, ok so far {ok}", flush=True)

        # -- what the sequential driver broadcasts (the input grouping of a layer)
        box = [[["q_proj", "k_proj", "v_proj"], ["o_proj"]]]
        dist.broadcast_object_list(box, src=0)
        ok &= box[0] == [["q_proj", "k_proj", "v_proj"], ["o_proj"]]
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    print(f"[rccl] one-rank worker: {'ok' if ok else 'FAILED'}", flush=True)
    sys.exit(0 if ok else 3)


if __name__ == "__main__":
    main()
