"""Host logic of the batched chains: which Linear groups of a layer go through the chain together
(``engine.gptq_linear.batchable``; GPU side: tests/test_gpu_batched_chains.py)."""
import torch


def test_batchable_splits_by_in_features_and_keeps_one_ragged_group_per_batch():
    from quantool_amd.engine.gptq_linear import batchable

    class Acc:
        def __init__(self, K):
            self.K = K

    def grp(K, r):
        return ([torch.empty((r, K), device="meta")], Acc(K))

    groups = [grp(512, 128), grp(256, 64), grp(512, 100), grp(512, 72), grp(256, 128), grp(512, 256)]
    assert batchable(groups) == [[0, 5, 2], [3], [4, 1]]
    many = [grp(128, 128) for _ in range(19)]
    assert [len(b) for b in batchable(many)] == [16, 3]


def test_scaled_shapes_keep_the_group_structure():
    from quantool_amd.engine.model_shapes import MODEL_SHAPES, scaled

    for name in ("llama-3-70b", "mixtral-8x7b"):
        big, small = MODEL_SHAPES[name], scaled(MODEL_SHAPES[name], 16)
        assert [g for g, _, _ in big.groups] == [g for g, _, _ in small.groups]
        assert [[n for n, _ in lins] for _, _, lins in big.groups] == [[n for n, _ in lins] for _, _, lins in small.groups]
        for (_, K, lins), (_, k, slins) in zip(big.groups, small.groups):
            assert k == max(128, K // 16 // 128 * 128) and k % 128 == 0
            assert all(r % 128 == 0 and r >= 128 for _, r in slins)
    assert scaled(MODEL_SHAPES["llama-3-8b"], 1) is MODEL_SHAPES["llama-3-8b"]
