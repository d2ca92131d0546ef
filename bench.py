#!/usr/bin/env python3
"""bench.py -- patches/s of the EfficientNet-B0 feature-extraction hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of 256 synthetic 224x224 u8 patches per
GPU (BASELINE.json configs[1]: "EfficientNet-B0 forward, batch=256 random 224x224 patches"),
resident in HBM before the timed region; for N > 1 ranks hold independent shards (weak scaling):
every step's (256, 1280) features land in the rank's block of G*256 rows and the block is
all-gathered over RCCL ONCE per G steps -- G = K by default, i.e. one gather of the rank's whole
(K*256, 1280) block at the end, which is what configs[3] describes (one gather of (n/N, 1280)
blocks); `--gather-every G` with G < K issues the gathers asynchronously behind the next group's
kernels (double-buffered blocks; MMC_BENCH_OVERLAP=0: synchronous).  Gather buffers are allocated
once (dist.FeatureGatherer).
Steps rotate over NBUF distinct input batches (308 MB > the 256 MB Infinity Cache), so the
compulsory input bytes of a step come from HBM, as they would in a real run.
Rank 0 prints ONE JSON line.  The oracle is imported only for the cpu_baseline leg.

`--gpus N` with N > 1 from a plain `python bench.py` (no WORLD_SIZE in the environment) launches
the N ranks itself: N child processes of this script with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set, started BEFORE this process touches torch or the GPU; rank 0's
JSON line is passed through and the exit status is non-zero when any rank fails (the reference's
fan-out: scripts/launch_processing.py:59-66, 199-233).  `--dry-run` exercises exactly that
plumbing -- launcher, rendezvous, barrier / MAX-over-ranks timing, ragged all-gather -- on the
gloo backend with rank-tagged blocks instead of features: no GPU, no backbone, `value` null.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from collections import defaultdict
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

BATCH = 256
NBUF = 8      # distinct input batches the steps rotate over: 8 x 38.5 MB > 256 MB of Infinity Cache


def synth_weights():
    """Synthetic seeded backbone weights (no checkpoint exists offline): the same generator the
    tests use, with the committed calibrated BN statistics."""
    from mermaid_classifier_amd.synthetic import synthetic_state_dict
    stats = dict(np.load(ROOT / "tests" / "golden" / "synth_bn_stats.npz"))
    return synthetic_state_dict(seed=0, bn_stats=stats)


def usable_cores() -> int:
    """Host cores this process may actually use (affinity mask and cgroup CPU quota), not os.cpu_count()."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                    n = min(n, max(1, q // period))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline(sd, budget_s: float = 15.0):
    """The reference's CPU path restated (oracle): config-1 geometry, one 4872x5568 image x 10
    points, crop (reflect pad) + transform + B0 forward (batch 10) + tolist, fp32, all host cores."""
    import torch
    from oracle import efficientnet_b0_ref as ref, pyspacer_ref
    cores = usable_cores()
    torch.set_num_threads(cores)
    net = ref.EfficientNetB0Ref({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    rng = np.random.default_rng(42)
    image = rng.integers(0, 255, (4872, 5568, 3), dtype=np.uint8)
    rows, cols = [812, 1624], [928, 1856, 2784, 3712, 4640]
    rowcols = [(r, c) for r in rows for c in cols][:10]
    pyspacer_ref.extract(net, image, rowcols, batch_size=10)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        feats = pyspacer_ref.extract(net, image, rowcols, batch_size=10)
        feats.tolist()
        n += len(rowcols)
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 2000:
            break
    return {"value": n / dt, "unit": "patches/s", "cores": cores, "kind": "port",
            "sample": f"{n // 10} x (1 image 4872x5568 x 10 points: crop+transform+B0 fp32 batch 10+tolist), torch CPU {torch.get_num_threads()} threads"}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of this process, which has not
    imported torch or touched the GPU (and never does).  Rank 0 inherits stdout (the JSON line); the other ranks'
    stdout goes to stderr.  Returns the exit status: 0 only if every rank exited 0."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    status = 0
    try:
        live = dict(enumerate(procs))
        while live:                    # poll every rank: a dead rank leaves the others waiting in a collective (minutes under RCCL)
            for r, p in list(live.items()):
                rc = p.poll()
                if rc is None:
                    continue
                del live[r]
                if rc != 0:
                    print(f"bench.py: rank {r} exited with status {rc}", file=sys.stderr)
                    status = status or (rc if rc > 0 else 1)
                    for q in live.values():
                        q.terminate()
            if live:
                time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return status


def gather_desc(world: int, K: int, G: int, overlap: bool, rows: int = BATCH, d: int = 1280) -> str:
    """Names the gather granularity in config.workload."""
    n = -(-K // G)
    return (f"{n} RCCL all-gather{'s' if n > 1 else ''} of ({G * rows}, {d}) fp32 rank blocks (one per {G} steps"
            + (", issued asynchronously behind the next group's kernels" if overlap else "") + f") x{world} ranks")


def dry_run(args, json_out) -> None:
    """The multi-rank plumbing without a GPU: gloo rendezvous, the step loop's barrier / MAX-over-ranks timing and the
    ragged all-gather of dist.gather_features on rank-tagged blocks.  Measures nothing: `value` is null."""
    import torch
    import torch.distributed as dist
    from mermaid_classifier_amd.dist import gather_features, shard_range
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("MMC_BENCH_DRY_FAIL_RANK") == str(rank):      # test hook: a rank that dies must fail the whole run
        os._exit(3)
    from mermaid_classifier_amd.dist import FeatureGatherer
    rows = 5                                     # stand-in for the 256 patches of a step
    K = max(args.steps, 1)
    G = K if args.gather_every <= 0 else min(args.gather_every, K)
    overlap = os.environ.get("MMC_BENCH_OVERLAP", "1") == "1" and G < K

    def run_groups(steps):
        """The step loop of the real run on rank-tagged rows: step s fills rows of the rank's group block, every G steps (and
        at the end, ragged) the block is gathered once.  Returns the gathered matrices in order."""
        outs, pend = [], []
        s0 = 0
        while s0 < steps:
            g = min(G, steps - s0)
            n_total = world * g * rows + (world - 1)          # ragged on purpose: the first world-1 ranks hold one row more
            lo, hi = shard_range(n_total, rank, world)
            block = torch.empty((hi - lo, 8), dtype=torch.float32)
            for i in range(hi - lo):                          # "extract": row = its global patch index (+ the group's offset)
                block[i] = float(lo + i + 1000 * (s0 // G))
            gat = FeatureGatherer(n_total, 8, block)
            if overlap:
                work, fin = gat.gather(block, async_op=True)
                pend.append((work, fin, n_total, s0 // G))
            else:
                outs.append((gat.gather(block), n_total, s0 // G))
            s0 += g
        for work, fin, n_total, gi in pend:
            work.wait()
            outs.append((fin(), n_total, gi))
        return outs

    for _ in range(args.warmup):
        gather_features(torch.zeros((shard_range(world + 1, rank, world)[1] - shard_range(world + 1, rank, world)[0], 8)), world + 1)
    dist.barrier()
    t0 = time.perf_counter()
    outs = run_groups(K)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = len(outs) == -(-K // G)
    for full, n_total, gi in outs:
        ok = ok and full.shape == (n_total, 8) and bool(torch.equal(full[:, 0], torch.arange(n_total, dtype=torch.float32) + 1000 * gi))
    if rank == 0:
        print(json.dumps({"metric": "patches/sec (224x224 EfficientNet-B0)", "value": None, "unit": "patches/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.item()) / max(args.steps, 1) * 1e3,
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "none (dry run)",
                          "config": {"workload": "dry run: launcher + gloo rendezvous + ragged all-gather of rank-tagged blocks, no GPU work; "
                                                 + gather_desc(world, K, G, overlap, rows, 8),
                                     "gather_every_steps": G, "gathers_per_run": -(-K // G), "gather_overlap": overlap,
                                     "parallelism": f"patch-sharded x{world}"}, "dry_run": True, "gather_ok": ok}),
              file=json_out, flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit("dry run: gathered matrix is not in global patch order")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-passes", type=int, default=5)
    ap.add_argument("--gather-every", type=int, default=0,
                    help="N > 1: all-gather the rank block once per this many steps (0 = once, at the end, over all K steps' rows)")
    ap.add_argument("--spread-blocks", type=int, default=3, help="timed blocks of --steps for the spread (the first one is `value`)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / gather plumbing on gloo, no GPU work")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))     # nothing above has imported torch or touched the GPU

    # The JSON line must be the only thing on stdout: RCCL prints a five-line version banner to fd 1 when a communicator is
    # created (N > 1).  Keep the real stdout aside for the JSON line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, json_out)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from mermaid_classifier_amd.backbone import Backbone
    from mermaid_classifier_amd import schedule

    sd = synth_weights()
    bb = Backbone(sd, device=local_rank, max_batch=BATCH)
    rng = np.random.default_rng(42 + rank)
    pbufs = [torch.from_numpy(rng.integers(0, 255, (BATCH, 224, 224, 3), dtype=np.uint8)).to(dev) for _ in range(NBUF)]
    patches = pbufs[0]
    feats = torch.empty((BATCH, 1280), dtype=torch.float32, device=dev)

    from mermaid_classifier_amd.dist import FeatureGatherer

    # N > 1 (configs[3]): a step's features go into the rank's block of G * 256 rows; the block is all-gathered ONCE per G steps,
    # G = K by default (one gather of the rank's whole block at the end).  With G < K the gather of group g is issued
    # asynchronously on RCCL's stream while group g + 1 computes (two blocks / two receive matrices, a block is refilled only after
    # the gather that read it has been waited for); MMC_BENCH_OVERLAP=0 makes it synchronous.  Every buffer is allocated here, once.
    use_dist = world > 1 or os.environ.get("MMC_BENCH_FORCE_DIST") == "1"
    if use_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    K = args.steps
    G = K if args.gather_every <= 0 else min(args.gather_every, K)
    overlap = use_dist and G < K and os.environ.get("MMC_BENCH_OVERLAP", "1") == "1"
    nblk = 2 if overlap else 1
    blocks = [torch.empty((G * BATCH, 1280), dtype=torch.float32, device=dev) for _ in range(nblk)] if use_dist else []
    gats = [FeatureGatherer(world * G * BATCH, 1280, blocks[0]) for _ in range(nblk)] if use_dist else []
    tailg = None                                   # the last, shorter group when G does not divide K
    if use_dist and K % G:
        tailg = FeatureGatherer(world * (K % G) * BATCH, 1280, blocks[0])
    pending = [None] * nblk
    state = {"in": 0, "s": 0}

    def wait_block(i):
        if pending[i] is not None:
            pending[i].wait()          # stream-level wait for the collective that read blocks[i]
            pending[i] = None

    def step():
        src = pbufs[state["in"] % NBUF]    # a fresh batch every step: input bytes come from HBM, not from the Infinity Cache
        state["in"] += 1
        bb.extract(src, out=feats)
        if not use_dist:
            return
        s = state["s"]
        gi, k = divmod(s, G)
        i = gi % nblk
        if k == 0:
            wait_block(i)
        blocks[i][k * BATCH:(k + 1) * BATCH].copy_(feats, non_blocking=True)     # same stream, right behind the pass
        state["s"] = s + 1
        last = s + 1 == K
        if k + 1 == G or last:
            g = gats[i] if k + 1 == G else tailg
            send = blocks[i] if k + 1 == G else blocks[i][: (k + 1) * BATCH]
            if overlap and not last:
                pending[i], _ = g.gather(send, async_op=True)
            else:
                g.gather(send)
        if last:
            state["s"] = 0

    def drain():
        for i in range(nblk):
            wait_block(i)

    # Engine initialisation, before the W warm-up steps: the library captures a pass into a HIP graph the third time the same
    # buffers come in (mmc_api.cpp run_pass) -- a one-off cost of a few ms that belongs to set-up like the weights upload, so
    # that a small W cannot push it into the timed region.  The RCCL communicator and every gather buffer are warmed the same
    # way (the first collective creates the communicator: hundreds of ms).
    for _ in range(3 * NBUF):          # every (input, output) combination seen three times: its graph exists
        bb.extract(pbufs[state["in"] % NBUF], out=feats)
        state["in"] += 1
    if use_dist:
        for g, blk in zip(gats, blocks):
            g.gather(blk)
        if tailg is not None:
            tailg.gather(blocks[0][: (K % G) * BATCH])
    for _ in range(args.warmup):
        bb.extract(pbufs[state["in"] % NBUF], out=feats)
        state["in"] += 1

    def timed_block():
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # `value` comes from the FIRST block of exactly K steps (the contract); the further blocks only feed `spread`.
    times = [timed_block() for _ in range(max(1, args.spread_blocks))]
    elapsed = times[0]
    if use_dist:
        lastg = tailg if tailg is not None else gats[((K - 1) // G) % nblk]
        nrows = ((K % G) or G) * BATCH
        mine = lastg.out[rank * nrows:(rank + 1) * nrows]
        if not torch.equal(mine[-BATCH:], feats):
            raise SystemExit("gathered feature block differs from the local features")
    if not np.isfinite(feats.float().sum().item()):
        raise SystemExit("non-finite features")

    if rank == 0:
        value = world * BATCH * args.steps / elapsed
        vals = [world * BATCH * args.steps / t for t in times]
        # per-kernel durations, HIP events on the launch stream, same resident workload
        per_kernel = defaultdict(lambda: [0.0, 0, 0, 0])  # ms, launches, algorithmic bytes, algorithmic flops
        sub = -(-BATCH // bb.lanes)   # each launch of the schedule covers one lane's sub-batch
        alg = {l.name: l for l in schedule.b0_launches(sub)}
        launched = []
        for _ in range(args.profile_passes):
            for name, ms in bb.profile(patches, feats):
                layer, kern = name.split("|")
                launched.append(layer)
                e = per_kernel[kern]
                e[0] += ms
                e[1] += 1
                e[2] += alg[layer].bytes
                e[3] += alg[layer].flops
        dom, (ms, launches, nbytes, nflops) = max(per_kernel.items(), key=lambda kv: kv[1][0])
        # which roof bounds the dominant kernel: the one its algorithmic work needs longer on
        mfma_bound = nflops / (schedule.MFMA_F16_PEAK_TFLOPS * 1e12) > nbytes / (schedule.HBM_PEAK_GBS * 1e9)
        if mfma_bound:
            achieved, peak, unit = nflops / (ms * 1e-3) / 1e12, schedule.MFMA_F16_PEAK_TFLOPS, "TFLOP/s"
        else:
            achieved, peak, unit = nbytes / (ms * 1e-3) / 1e9, schedule.HBM_PEAK_GBS, "GB/s"
        tot = schedule.totals(BATCH, launched)
        # HBM traffic of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # in separate runs, FETCH_SIZE doubled per the gfx950 correction; tools/summarize_profiles.py), if present
        traffic, traffic_source = None, None
        try:
            pmc = sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"))[-1]
            table = json.loads(pmc.read_text())
            key = dom.split("<")[0]
            targs = lambda name: [t.strip() for t in name.split("<")[1].split(">")[0].split(",")][:3]   # noqa: E731
            hits = [v for k, v in table.items() if k.split("<")[0].replace("_kernel", "") == key.replace("_kernel", "")
                    and (("<" not in dom) or ("<" in k and targs(k) == targs(dom)))]
            if hits:
                traffic = sum(h["read_bytes_per_launch"] + h["write_bytes_per_launch"] for h in hits) / len(hits)
                traffic_source = f"profiles/{pmc.name} (committed rocprofv3 --pmc passes of this command, not measured in this run)"
        except Exception:
            traffic, traffic_source = None, None
        roofline = {"bound": "mfma" if mfma_bound else "hbm", "kernel": dom, "achieved": achieved, "peak": peak, "unit": unit,
                    "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_source,
                    "avg_launch_us": ms / launches * 1e3, "launches_per_step": launches // args.profile_passes,
                    "patches_per_launch": sub,
                    # the lanes run this kernel side by side (kernel trace: in lockstep), each launch on its share of the CUs, while
                    # `peak` is the whole chip's: the chip as a whole is at lanes x frac of that peak while the kernel runs
                    "launches_in_flight": bb.lanes, "frac_all_launches_in_flight": bb.lanes * achieved / peak,
                    "alg_bytes_per_launch": nbytes / launches, "alg_flops_per_launch": nflops / launches,
                    "whole_net": {"alg_GB_per_step": tot["bytes"] / 1e9,
                                  "hbm_frac_at_value": tot["bytes_per_patch"] * value / 1e9 / schedule.HBM_PEAK_GBS,
                                  "mfma_frac_at_value": tot["flops_per_patch"] * value / 1e12 / schedule.MFMA_F16_PEAK_TFLOPS}}
        out = {
            "metric": "patches/sec (224x224 EfficientNet-B0)", "value": value, "unit": "patches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "spread": {"unit": "patches/s", "blocks": len(times), "steps_per_block": K,
                       "min": min(vals), "median": float(np.median(vals)), "max": max(vals)},
            "config": {"workload": "EfficientNet-B0 forward, batch=256 random 224x224 u8 patches per GPU -> (256,1280) fp32"
                                   + ("; " + gather_desc(world, K, G, overlap) if use_dist else ""),
                       "per_gpu_batch": BATCH, "global_batch": world * BATCH, "weights": "synthetic seed 0",
                       "input_batches_rotated": NBUF,
                       "lanes_per_gpu": bb.lanes,
                       **({"gather_every_steps": G, "gathers_per_run": -(-K // G), "gather_overlap": overlap} if use_dist else {}),
                       "parallelism": f"patch-sharded x{world}"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd)
        kernels = sorted(((k, v[0] / args.profile_passes) for k, v in per_kernel.items()), key=lambda kv: -kv[1])
        print("# per-kernel ms/step (HIP events): " + ", ".join(f"{k}={v:.3f}" for k, v in kernels), file=sys.stderr)
        print(json.dumps(out), file=json_out, flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
