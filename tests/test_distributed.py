"""world_size-2 gloo tests (CPU): the batch-sharded multi-GPU path and its one exchange
step, the all-gather of per-image features.  Acceptance: the bank built
after the all-gather is bit-identical to the single-rank bank, and the gradient that flows
back through the gather (the image encoder is trained in the reference) equals the
single-rank gradient of every image, on the rank that owns it."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "mm-interleaved_amd")]
    from mmfs_amd import bank
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)                     # same data on every rank
        num = torch.tensor([3, 1, 2, 0, 1])                      # images per sequence, 5 sequences
        n_img = int(num.sum())
        levels = [torch.randn(n_img, 6, s, s, generator=g) for s in (8, 4, 2)]
        packed = bank.pack_image_levels(levels)                  # [7, 84, 6] -- the single-rank truth
        want_bank = bank.llm_feature_bank(packed, num, 3)

        # each rank "encodes" only its block of the images ...
        i0, i1 = bank.local_image_range(n_img, rank, world)
        mine = packed[i0:i1].clone().requires_grad_(True)
        gathered = bank.all_gather_image_features(mine, n_img)
        assert gathered.requires_grad
        assert torch.equal(gathered.detach(), packed), "gathered features differ from the single-rank tensor"
        # ... and builds the bank of its own batch shard from the gathered tensor
        lo, hi = bank.shard_batch(num.numel(), rank, world)
        first = int(num[:lo].sum())
        local = bank.llm_feature_bank(gathered[first:first + int(num[lo:hi].sum())], num[lo:hi], 3)
        assert torch.equal(local.detach(), want_bank[lo:hi])
        # backward: every rank's loss touches images of other ranks; the gradient of an image must come
        # back to its owner, summed over the ranks that used it.  Truth: the single-rank bank with the
        # per-sequence weights every rank can recompute.
        wts = torch.randn(want_bank.shape, generator=g)
        (local * wts[lo:hi]).sum().backward()
        ref = packed.clone().requires_grad_(True)
        (bank.llm_feature_bank(ref, num, 3) * wts).sum().backward()
        assert torch.allclose(mine.grad, ref.grad[i0:i1], rtol=0, atol=1e-6), "gradient did not return to the owning rank"
        assert float(ref.grad[i0:i1].abs().max()) > 0
        # every sequence is processed by exactly one rank
        counts = torch.zeros(num.numel())
        counts[lo:hi] = 1
        dist.all_reduce(counts)
        assert torch.equal(counts, torch.ones(num.numel()))
        open(os.path.join(tmp, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def _worker_idle_rank(rank, world, port, tmp):
    """Rank 1's sequences show no image at all: its loss does not depend on the gathered tensor, and without
    ``keep_in_graph`` it would never enter the gather's backward collective."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "mm-interleaved_amd")]
    from mmfs_amd import bank
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1)
        num = torch.tensor([2, 1, 0, 0])                         # sequences 2 and 3 (rank 1's shard) have no image
        n_img = int(num.sum())
        packed = torch.randn(n_img, 20, 6, generator=g)
        i0, i1 = bank.local_image_range(n_img, rank, world)
        mine = packed[i0:i1].clone().requires_grad_(True)
        gathered = bank.all_gather_image_features(mine, n_img)
        lo, hi = bank.shard_batch(num.numel(), rank, world)
        first, cnt = int(num[:lo].sum()), int(num[lo:hi].sum())
        assert (cnt == 0) == (rank == 1)
        local = bank.llm_feature_bank(gathered[first:first + cnt], num[lo:hi], 2)
        wts = torch.randn(num.numel(), 2, 20, 6, generator=g)
        loss = (local * wts[lo:hi]).sum() + torch.zeros((), requires_grad=True).sum()
        bank.keep_in_graph(loss, gathered).backward()            # (plain ``loss.backward()`` would hang rank 0 here)
        ref = packed.clone().requires_grad_(True)
        (bank.llm_feature_bank(ref, num, 2) * wts).sum().backward()
        assert torch.allclose(mine.grad, ref.grad[i0:i1], rtol=0, atol=1e-6)
        open(os.path.join(tmp, f"idle_ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def _worker_ragged(rank, world, port, tmp, n_img, nums):
    """World sizes 4 and 8 with image counts that do not divide by the world size: the last rank(s) hold fewer images --
    or NONE -- and pad their block with zeros in the forward gather (bank.py: all_gather_image_features); sequences are
    sharded unevenly too; some ranks' shards show no image.  Same acceptance as the world-size-2 test."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "mm-interleaved_amd")]
    from mmfs_amd import bank
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(2)
        num = torch.tensor(nums)
        assert int(num.sum()) == n_img
        n_max = int(num.max())
        packed = torch.randn(n_img, 20, 6, generator=g)          # the single-rank truth
        packed[0, 0, 0] = float("inf")                          # a non-finite feature must not leak through keep_in_graph
        want_bank = bank.llm_feature_bank(packed, num, n_max)
        i0, i1 = bank.local_image_range(n_img, rank, world)
        per = bank.images_per_rank(n_img, world)
        assert 0 <= i1 - i0 <= per
        mine = packed[i0:i1].clone().requires_grad_(True)       # (may be EMPTY: a rank past the last image)
        gathered = bank.all_gather_image_features(mine, n_img)
        assert gathered.shape[0] == n_img
        assert torch.equal(gathered.detach(), packed), "gathered features differ from the single-rank tensor"
        lo, hi = bank.shard_batch(num.numel(), rank, world)
        first, cnt = int(num[:lo].sum()), int(num[lo:hi].sum())
        local = bank.llm_feature_bank(gathered[first:first + cnt], num[lo:hi], n_max)
        assert torch.equal(local.detach(), want_bank[lo:hi])
        wts = torch.randn(want_bank.shape, generator=g)
        wts[torch.isinf(want_bank)] = 0.0                        # (keep the truth's gradient finite where the feature is not)
        loss = (torch.nan_to_num(local, posinf=0.0) * wts[lo:hi]).sum() + torch.zeros((), requires_grad=True).sum()
        loss = bank.keep_in_graph(loss, gathered)
        assert bool(torch.isfinite(loss)), "keep_in_graph must not read the gathered values"
        loss.backward()
        ref = packed.clone().requires_grad_(True)
        (torch.nan_to_num(bank.llm_feature_bank(ref, num, n_max), posinf=0.0) * wts).sum().backward()
        if i1 > i0:
            assert torch.allclose(mine.grad, ref.grad[i0:i1], rtol=0, atol=1e-5), "gradient did not return to the owning rank"
        else:
            assert mine.grad is None or mine.grad.numel() == 0
        open(os.path.join(tmp, f"ragged_ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world, n_img, nums", [
    (4, 7, [3, 1, 0, 2, 0, 1]),                      # 7 images on 4 ranks: blocks of 2, the last rank holds ONE; 6 sequences on 4 ranks
    (4, 5, [2, 0, 0, 3, 0]),                         # blocks of 2: rank 2 holds one image, rank 3 NONE (all padding)
    (8, 11, [1, 2, 0, 3, 0, 0, 1, 2, 0, 2, 0]),      # blocks of 2: rank 5 holds one, ranks 6 and 7 none; 11 sequences on 8 ranks
    (8, 3, [0, 1, 0, 0, 2, 0, 0, 0, 0]),             # fewer images than ranks
])
def test_ragged_image_blocks_and_idle_ranks_at_world_4_and_8(tmp_path, world, n_img, nums):
    mp.spawn(_worker_ragged, args=(world, _free_port(), str(tmp_path), n_img, nums), nprocs=world, join=True)
    assert all((tmp_path / f"ragged_ok{r}").exists() for r in range(world))


def test_a_rank_without_images_still_joins_the_backward_collective(tmp_path):
    world = 2
    mp.spawn(_worker_idle_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"idle_ok{r}").exists() for r in range(world))


def test_feature_all_gather_and_batch_sharding_gloo(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_shard_batch_is_a_partition():
    from mmfs_amd.bank import shard_batch
    for n in (0, 1, 7, 8, 13):
        for w in (1, 2, 4, 8):
            spans = [shard_batch(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
