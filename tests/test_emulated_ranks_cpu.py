"""The "nccl" branches of mpreid/distributed.py with P > 1, on host tensors: P virtual ranks as threads of this process
(tests/emulated_group.py).  The index arithmetic that only a multi-rank RCCL run reaches -- the ragged split sizes of the
all_to_all_single in column_to_row_blocks, the padded gather of ragged blocks, the padded all-gathers -- is executed and
checked against the plain concatenation; the emulator asserts what RCCL requires (contiguity, dtypes, matching splits).
The GPU counterpart (device tensors, the evaluator and the sharded re-ranking end to end): tests/test_gpu_emulated_ranks.py."""
import numpy as np
import pytest
import torch

from emulated_group import EmulatedWorld
from mpreid import distributed as D


@pytest.mark.parametrize("world,nq,ng", [(2, 7, 11), (3, 10, 37), (8, 41, 643), (8, 5, 9), (5, 3, 4)])
def test_column_to_row_blocks_and_host_concat_ragged(world, nq, ng):
    rng = np.random.default_rng(world * 1000 + nq)
    full = torch.from_numpy(rng.standard_normal((nq, ng)).astype(np.float32))
    ng_sizes = D.shard_sizes(ng, world)
    W = EmulatedWorld(world, require_cuda=False)

    def rank_fn(r):
        g_lo, g_hi = D.shard_range(ng, r, world)
        q_lo, q_hi = D.shard_range(nq, r, world)
        block = full[:, g_lo:g_hi].contiguous()
        rows = D.column_to_row_blocks(block, nq, ng_sizes)                 # all_to_all_single, ragged both ways
        assert torch.equal(rows, full[q_lo:q_hi]), r
        cat = D.gather_column_blocks_to_host(block, dst=0)                  # padded gather of ragged column blocks
        rcat = D.gather_row_blocks_to_host(rows, dst=0)                     # ... and of ragged row blocks
        gathered = D.all_gather_rows(full[q_lo:q_hi].contiguous(), nq)      # padded all_gather_into_tensor
        assert torch.equal(gathered, full)
        h16 = D.all_gather_rows(full[q_lo:q_hi].to(torch.float16).contiguous().view(torch.int16), nq)   # fp16 bits: byte view
        assert h16.dtype == torch.int16 and torch.equal(h16.view(torch.float16), full.to(torch.float16))
        sizes = [3 * s + (1 if i % 2 else 0) for i, s in enumerate(D.shard_sizes(nq, world))]
        mine = torch.arange(sizes[r], dtype=torch.int32) + 1000 * r
        pieces = D.all_gather_ragged(mine, sizes)                           # the index pieces of the sharded re-ranking
        for i, pc in enumerate(pieces):
            assert torch.equal(pc, torch.arange(sizes[i], dtype=torch.int32) + 1000 * i)
        return cat, rcat

    res = W.run(rank_fn)
    assert np.array_equal(res[0][0], full.numpy()) and np.array_equal(res[0][1], full.numpy())
    assert all(r[0] is None and r[1] is None for r in res[1:])
    assert "all_to_all_single" in W.log and "gather" in W.log and "all_gather_into_tensor" in W.log


def test_emulator_rejects_what_rccl_would_not_survive():
    """mismatched all_to_all_single splits, a non-contiguous buffer, a dtype NCCL does not have: the emulator raises"""
    W = EmulatedWorld(2, require_cuda=False)

    def bad_splits(r):
        send = torch.zeros(4)
        recv = torch.zeros(4)
        D._pg().all_to_all_single(recv, send, output_split_sizes=[2, 2], input_split_sizes=[1, 3] if r == 0 else [2, 2])
    with pytest.raises(AssertionError):
        W.run(bad_splits)
    with pytest.raises(AssertionError):
        EmulatedWorld(2, require_cuda=False).run(lambda r: D._pg().all_gather_into_tensor(torch.zeros(4, 2), torch.zeros(2, 4).t()))
    with pytest.raises(AssertionError):
        EmulatedWorld(2, require_cuda=False).run(lambda r: D._pg().all_gather_into_tensor(torch.zeros(4, dtype=torch.int16),
                                                                                        torch.zeros(2, dtype=torch.int16)))
