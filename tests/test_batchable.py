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


def test_batched_entry_refuses_what_batchable_would_not_have_built():
    """Argument errors of ``gptq_quantize_batched`` are raised on the host before any device call."""
    import pytest

    from quantool_amd.engine.gptq_linear import gptq_quantize_batched
    from quantool_amd.engine.schemes import QuantArgs

    class Acc:
        def __init__(self, K, n=4):
            self.K, self.n = K, n
            self.G = torch.empty((K, K), device="meta")

    def grp(K, r, n=4):
        return ([torch.empty((r, K), device="meta")], Acc(K, n))

    qa = QuantArgs()
    assert gptq_quantize_batched([], qa) == []
    with pytest.raises(ValueError, match="share in_features"):
        gptq_quantize_batched([grp(256, 128), grp(512, 128)], qa)
    with pytest.raises(ValueError, match="multiple of 128"):
        gptq_quantize_batched([grp(256, 100), grp(256, 128)], qa)          # the ragged group must go last
    with pytest.raises(ValueError, match="no calibration samples"):
        gptq_quantize_batched([grp(256, 128, n=0)], qa)
    with pytest.raises(ValueError, match="at most"):
        gptq_quantize_batched([grp(256, 128) for _ in range(17)], qa)
    with pytest.raises(ValueError, match="block_size"):
        gptq_quantize_batched([grp(256, 128)], qa, block_size=64)
    with pytest.raises(ValueError, match="does not match in_features"):
        gptq_quantize_batched([([torch.empty((128, 200), device="meta")], Acc(256))], qa)


def test_stack_batches_groups_equal_shapes_up_to_the_token_budget():
    """First-layer capture of the sequential driver: consecutive equal-shape samples share a forward."""
    from quantool_amd.engine.sequential import stack_batches

    bs = ([{"input_ids": torch.arange(6).reshape(1, 6) + i} for i in range(5)] + [{"input_ids": torch.zeros(1, 4, dtype=torch.long)}]
          + [{"input_ids": torch.ones(1, 6, dtype=torch.long)}])
    out = stack_batches(bs, 12)
    assert [tuple(b["input_ids"].shape) for b in out] == [(2, 6), (2, 6), (1, 6), (1, 4), (1, 6)]
    assert torch.equal(torch.cat([b["input_ids"] for b in out if b["input_ids"].shape[1] == 6])[:5],
                       torch.cat([b["input_ids"] for b in bs[:5]]))
    assert len(stack_batches(bs, 0)) == len(bs)                                    # QT_CALIB_BATCH_TOKENS=0: one sample per forward
    odd = [{"input_ids": torch.zeros(1, 6, dtype=torch.long), "extra": 3}, {"input_ids": torch.zeros(1, 6, dtype=torch.long), "extra": 3}]
    assert len(stack_batches(odd, 64)) == 2                                        # non-tensor entries: left alone
