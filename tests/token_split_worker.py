"""One rank of the partitioning-B rehearsal (see test_gpu_token_split.py): both ranks share the box's
single GPU, gloo stands in for RCCL.  argv: rank world port actorder symmetric"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    actorder = None if sys.argv[4] == "none" else sys.argv[4]
    symmetric = sys.argv[5] == "1"
    from quantool_amd.engine.gptq_linear import HessianAccumulator, gptq_quantize_shared
    from quantool_amd.engine.schemes import QuantArgs
    from quantool_amd.engine.sharding import gptq_quantize_token_split

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        K = 384
        g = torch.Generator().manual_seed(11)
        X = torch.randn((6, 100, K), generator=g).to(torch.bfloat16).to(dev)       # 6 calibration samples
        X[..., 5] *= 8.0
        # 37 rows: uneven split; 1 row: the second rank owns nothing of it
        Ws = [(torch.randn((r, K), generator=g) * 0.02).to(torch.bfloat16).to(dev) for r in (64, 37, 1)]
        qa = QuantArgs(num_bits=4, symmetric=symmetric, group_size=128, actorder=actorder)
        got = gptq_quantize_token_split(Ws, [X[i:i + 1] for i in range(rank, 6, world)], qa)
        torch.cuda.synchronize()
        ok = True
        if rank == 0:
            # single-process reference on a Gram matrix summed in the same order: G0 + G1 (+ ...)
            accs = []
            for r in range(world):
                a = HessianAccumulator(K, dev)
                for i in range(r, 6, world):
                    a.add(X[i:i + 1])
                accs.append(a)
            tot = HessianAccumulator(K, dev)
            tot.G.copy_(accs[0].G)
            for a in accs[1:]:
                tot.G += a.G
            tot.n = sum(a.n for a in accs)
            assert tot.n == 6
            want = gptq_quantize_shared(Ws, tot, qa)
            for w, parts, ref in zip(Ws, got, want):
                ok &= torch.equal(parts["weight_packed"], ref.weight_packed)
                ok &= torch.equal(parts["weight_scale"], ref.weight_scale)
                ok &= parts["weight_packed"].shape[0] == w.shape[0]
                if not symmetric:
                    ok &= torch.equal(parts["weight_zero_point"], ref.weight_zero_point)
                else:
                    ok &= "weight_zero_point" not in parts
                if actorder == "group":
                    ok &= torch.equal(parts["weight_g_idx"], ref.weight_g_idx)
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank != 0:
            # every rank holds the full result: compare a checksum with rank 0's
            pass
        sums = [None] * world
        dist.all_gather_object(sums, [int(p["weight_packed"].to(torch.int64).sum().item()) for p in got])
        ok &= all(s == sums[0] for s in sums)
        sys.exit(0 if (ok and int(flag.item()) == 1) else 3)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
