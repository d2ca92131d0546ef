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


def _run_bench(args, extra_env=None):
    import subprocess
    import sys
    from pathlib import Path
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "bench.py")] + args,
                          env=env, capture_output=True, text=True, timeout=300)


def test_bench_launches_its_own_ranks_from_a_plain_invocation():
    """`python bench.py --gpus 2` with no launcher (no WORLD_SIZE): the script starts the two ranks itself before touching
    torch, rank 0's single JSON line comes through, exit status 0.  --dry-run swaps RCCL/GPU work for gloo + rank-tagged
    blocks, so this is the launcher, rendezvous, timing and ragged-gather plumbing of the N > 1 path (the reference's
    fan-out: scripts/launch_processing.py:59-66, 199-233)."""
    import json
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["gather_ok"] is True and out["value"] is None
    assert out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    # configs[3]: ONE gather of the rank's whole block (all K steps' rows) by default, and the line says so
    cfg = out["config"]
    assert cfg["gather_every_steps"] == 3 and cfg["gathers_per_run"] == 1 and cfg["gather_overlap"] is False
    assert "1 RCCL all-gather of (15, 8)" in cfg["workload"]


def test_bench_gathers_once_per_group_with_a_ragged_last_group():
    """--gather-every 2 over 5 steps: groups of 2, 2 and 1 steps (ragged K), gathered asynchronously behind the next group; the
    dry run checks every gathered matrix for global row order (ragged shards inside each group as well)."""
    import json
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "5", "--warmup", "0", "--gather-every", "2"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    cfg = out["config"]
    assert out["gather_ok"] is True and cfg["gather_every_steps"] == 2 and cfg["gathers_per_run"] == 3 and cfg["gather_overlap"] is True
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "5", "--warmup", "0", "--gather-every", "2"], {"MMC_BENCH_OVERLAP": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["config"]["gather_overlap"] is False


def _gatherer_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mermaid_classifier_amd.dist import FeatureGatherer, shard_range
        res = []
        for n_total in (8, 7):                      # even and ragged shards; the SAME gatherer is reused three times
            lo, hi = shard_range(n_total, rank, world)
            g = FeatureGatherer(n_total, 4, torch.zeros(1))
            ptr = g.out.data_ptr()
            for it in range(3):
                local = (torch.arange(lo, hi, dtype=torch.float32) + 100 * it).view(-1, 1).repeat(1, 4)
                if it == 1:
                    work, fin = g.gather(local, async_op=True)
                    work.wait()
                    full = fin()
                else:
                    full = g.gather(local)
                res.append((n_total, it, full[:, 0].clone().numpy(), g.out.data_ptr() == ptr))
            try:
                g.gather(torch.zeros((hi - lo + 1, 4)))
                res.append("no error")
            except ValueError:
                pass
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_feature_gatherer_reuses_its_buffers_and_keeps_patch_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gatherer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in results:
        assert "no error" not in res
        for n_total, it, col, same_buf in res:
            np.testing.assert_array_equal(col, np.arange(n_total) + 100 * it)
            assert same_buf          # no allocation per call


def test_bench_self_launch_fails_when_a_rank_dies():
    r = _run_bench(["--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"], {"MMC_BENCH_DRY_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "rank 1 exited with status" in r.stderr


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    r = _run_bench(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
def test_two_rank_rccl_gather_is_in_patch_order():
    """gather_features over RCCL with a ragged n_total, one rank per GPU -- needs two GPUs (the 1-GPU test box skips it; RCCL
    refuses two ranks on one device)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, 2, port, 11, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, col0 in results:
        np.testing.assert_array_equal(col0, np.arange(11))


def _nccl_worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        from mermaid_classifier_amd.dist import gather_features, shard_range
        lo, hi = shard_range(n_total, rank, world)
        local = torch.arange(lo, hi, dtype=torch.float32, device="cuda").view(-1, 1).repeat(1, 4)
        out = gather_features(local, n_total)
        q.put((rank, out[:, 0].cpu().numpy()))
    finally:
        dist.destroy_process_group()


# ---- the same exchange through the C ABI alone (include/mmc.h: mmc_dist_* / mmc_gather_features) ----

def test_native_gatherer_rejects_a_malformed_id_without_touching_rccl():
    from mermaid_classifier_amd.dist import NativeGatherer
    with pytest.raises(ValueError, match="128 bytes"):
        NativeGatherer(b"short", 0, 1)


def test_c_abi_gather_rejects_bad_arguments():
    from mermaid_classifier_amd import _lib
    lib = _lib.lib()
    assert lib.mmc_dist_unique_id(None) == _lib.MMC_ERR_ARG
    assert lib.mmc_gather_features(None, None, 0, 1280, None, None, None) == _lib.MMC_ERR_ARG
    assert b"bad argument" in lib.mmc_last_error()
    lib.mmc_dist_destroy(None)   # a no-op, as free(NULL)


@pytest.mark.gpu
def test_c_abi_gather_world_of_one_on_the_device():
    """mmc_dist_unique_id -> mmc_dist_create -> mmc_gather_features with one rank: RCCL is resolved by the library itself
    (no torch.distributed), the block comes back unchanged, counts-less and with explicit counts."""
    from mermaid_classifier_amd import _lib
    from mermaid_classifier_amd.dist import NativeGatherer
    import ctypes as C
    g = NativeGatherer(NativeGatherer.unique_id(), 0, 1, device=0)
    try:
        local = torch.arange(7 * 1280, dtype=torch.float32, device="cuda").view(7, 1280)
        out = g.gather(local, 7)
        torch.cuda.synchronize()
        assert out.data_ptr() != local.data_ptr()
        np.testing.assert_array_equal(out.cpu().numpy(), local.cpu().numpy())
        out2 = torch.zeros_like(local)
        counts = (C.c_int64 * 1)(7)
        _lib.check(_lib.lib().mmc_gather_features(g._h, local.data_ptr(), 7, 1280, counts, out2.data_ptr(), None))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out2.cpu().numpy(), local.cpu().numpy())
        counts[0] = 6   # a block height that contradicts n_local is refused before anything is sent
        assert _lib.lib().mmc_gather_features(g._h, local.data_ptr(), 7, 1280, counts, out2.data_ptr(), None) == _lib.MMC_ERR_ARG
    finally:
        g.close()


@pytest.mark.gpu
def test_c_abi_gather_two_ranks_ragged():
    """Two processes, one GPU each, the id handed over through a file: needs two GPUs (the 1-GPU test box skips it)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    import tempfile
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as td:
        procs = [ctx.Process(target=_native_worker, args=(r, 2, os.path.join(td, "id"), 11, q)) for r in range(2)]
        for p in procs:
            p.start()
        results = [q.get(timeout=300) for _ in range(2)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    for rank, col0 in results:
        np.testing.assert_array_equal(col0, np.arange(11))


def _native_worker(rank, world, id_path, n_total, q):
    import time
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from mermaid_classifier_amd.dist import NativeGatherer, shard_range
    torch.cuda.set_device(rank)
    if rank == 0:
        with open(id_path + ".tmp", "wb") as f:
            f.write(NativeGatherer.unique_id())
        os.replace(id_path + ".tmp", id_path)
    while not os.path.exists(id_path):
        time.sleep(0.05)
    g = NativeGatherer(open(id_path, "rb").read(), rank, world, device=rank)
    lo, hi = shard_range(n_total, rank, world)
    local = torch.arange(lo, hi, dtype=torch.float32, device="cuda").view(-1, 1).repeat(1, 4)
    out = g.gather(local, n_total)
    torch.cuda.synchronize()
    q.put((rank, out[:, 0].cpu().numpy()))
    g.close()
