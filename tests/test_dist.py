"""CPU tests of the N>1 path: world_size-2 (and 3) gloo process groups exercise the same
shard_range / gather_features code that runs over RCCL on the GPUs (bench.py --gpus N)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mermaid_classifier_amd.dist import extract_sharded, shard_range
        patches = np.arange(n_total)          # stand-in for the patch list: "features" are a function of the index

        def fake_extract(block):
            idx = torch.as_tensor(np.asarray(block), dtype=torch.float32)
            return torch.stack([idx, idx * 2 + 1, torch.full_like(idx, float(rank))], dim=1) if len(block) else torch.zeros((0, 3))

        out = extract_sharded(fake_extract, patches)
        lo, hi = shard_range(n_total, rank, world)
        q.put((rank, out.numpy(), lo, hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 10), (2, 7), (3, 8), (2, 1), (2, 0)])
def test_gather_is_in_patch_order_for_even_and_ragged_shards(world, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, out, lo, hi in results:
        assert out.shape == (n_total, 3)
        np.testing.assert_array_equal(out[:, 0], np.arange(n_total))          # global patch order
        np.testing.assert_array_equal(out[:, 1], np.arange(n_total) * 2 + 1)
        owners = np.concatenate([np.full(b - a, r) for r in range(world)
                                 for a, b in [__import__("mermaid_classifier_amd.dist", fromlist=["x"]).shard_range(n_total, r, world)]]) if n_total else np.zeros(0)
        np.testing.assert_array_equal(out[:, 2], owners)                      # every row came from its owner


def test_shard_range_partitions_exactly():
    from mermaid_classifier_amd.dist import chunk_items, shard_range
    for n in (0, 1, 7, 8, 250_000, 1_000_000):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)
    # the reference's round-robin rule (tests/sagemaker_launcher/test_launch_processing.py:20-36)
    assert chunk_items(list(range(7)), 3) == [[0, 3, 6], [1, 4], [2, 5]]
    assert chunk_items([1, 2], 5) == [[1], [2]]
