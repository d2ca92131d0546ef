#!/usr/bin/env python3
"""Per-kernel SQ / TCC / TCP counter table from the PMC passes of tools/gpu.sh counters.

  python tools/summarize_counters.py <round-tag> <dir with sq1/ sq2/ tcc/ fetch/ write/> [title suffix] [profiled command + what a launch covers]

Writes profiles/<tag>_sq_counters.md.  Per launch (average over the launches of the run, kernels serialised by the
profiler).  SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles summed over waves, so they are reported as
fractions of SQ_WAVE_CYCLES; SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over SIMDs (16 per
v_mfma_f32_16x16x32_f16), so MFMA-busy = that / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs.
"""
import collections
import csv
import glob
import os
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def short(name):
    name = re.sub(r"\(.*", "", name)
    m = re.match(r"_Z\d+(\w+?)I(.*?)EEv", name)
    if m:
        args = re.findall(r"L[ib](\d+)E", m.group(2))
        return f"{m.group(1)}<{','.join(args)}>"
    m = re.match(r"_Z\d+([a-z0-9_]+?_kernel)", name)
    if m:
        return m.group(1)
    return name.replace("void ", "")[:48]


def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    dur = collections.defaultdict(lambda: [0.0, 0])
    files = glob.glob(f"{d}/*/*counter_collection.csv")
    if not files:
        return acc, cnt, dur
    seen = set()
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]] += 1
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            dur[k][1] += 1
    return acc, cnt, dur


def main():
    tag, base = sys.argv[1], sys.argv[2]
    suffix = sys.argv[3] if len(sys.argv) > 3 else ""
    what = sys.argv[4] if len(sys.argv) > 4 else ("python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-passes 1` (tools/gpu.sh counters; one counter "
                                                  "group per run; every launch covers one lane = 128 patches")
    passes = {p: load(f"{base}/{p}") for p in ("sq1", "sq2", "tcc", "fetch", "write")}

    def val(p, k, c):
        a, n, _ = passes[p]
        return a[k][c] / n[k][c] if n[k].get(c) else float("nan")

    kernels = sorted(passes["sq1"][2], key=lambda k: -passes["sq1"][2][k][0])
    kernels = [k for k in kernels if passes["sq1"][2][k][0] > 0 and "rocclr" not in k and val("sq1", k, "SQ_WAVES") > 0]
    out = [f"# SQ / TCC / TCP counters per kernel launch ({tag}{suffix})", "",
           "command per pass: `rocprofv3 --pmc <group> --output-format csv -- " + what +
           "; kernels are serialised by the profiler, so durations are those of a kernel alone on the chip)", "",
           "Groups: sq1 = SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY "
           "SQ_INSTS_VALU; sq2 = SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES "
           "SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS; tcc = TCP_TCC_READ_REQ_sum TCC_HIT_sum "
           "TCC_MISS_sum TCC_REQ_sum; fetch = FETCH_SIZE GRBM_GUI_ACTIVE; write = WRITE_SIZE.", "",
           "## Where the wave time goes (fractions of SQ_WAVE_CYCLES) and what the matrix pipe does", "",
           "`MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): share of all matrix-pipe cycles of the chip "
           "during the launch (a kernel with one workgroup per patch occupies 128 of 256 CUs: its pipes can reach 50 % at most); "
           "`coexec` = SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES; `trans` = transcendental share of VALU instructions; "
           "`VALU port` = 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): share of the chip's vector-issue cycles the "
           "launch uses (the per-wave fractions to its left shrink with the number of waves that share a SIMD).", "",
           "| kernel | launches | us | waves | VALU active | any inst active | wait any (s_waitcnt/barrier) | wait inst (issue stall) | "
           "VALU port | VALU insts/wave | trans | MFMA insts/wave | cycles per MFMA | MFMA busy | coexec | LDS conflict / LDS active |",
           "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for k in kernels:
        wc = val("sq1", k, "SQ_WAVE_CYCLES")
        waves = val("sq1", k, "SQ_WAVES")
        nm = val("sq2", k, "SQ_INSTS_MFMA")
        mb = val("sq2", k, "SQ_VALU_MFMA_BUSY_CYCLES")
        gui = val("fetch", k, "GRBM_GUI_ACTIVE")
        d = passes["sq1"][2][k]
        out.append(f"| `{k}` | {d[1]} | {d[0] / d[1]:.1f} | {waves:.0f} | {val('sq1', k, 'SQ_ACTIVE_INST_VALU') / wc:.2f} | "
                   f"{val('sq1', k, 'SQ_ACTIVE_INST_ANY') / wc:.2f} | {val('sq1', k, 'SQ_WAIT_ANY') / wc:.2f} | "
                   f"{val('sq1', k, 'SQ_WAIT_INST_ANY') / wc:.2f} | "
                   f"{(4 * val('sq1', k, 'SQ_ACTIVE_INST_VALU') / (gui / 8 * 1024) if gui == gui and gui else float('nan')):.2f} | "
                   f"{val('sq1', k, 'SQ_INSTS_VALU') / waves:.0f} | "
                   f"{val('sq2', k, 'SQ_INSTS_VALU_TRANS_F32') / max(1.0, val('sq1', k, 'SQ_INSTS_VALU')):.2f} | {nm / waves:.0f} | "
                   f"{(mb / nm if nm else float('nan')):.1f} | {(mb / (gui / 8 * 1024) if gui == gui and gui else float('nan')):.3f} | "
                   f"{(val('sq2', k, 'SQ_VALU_MFMA_COEXEC_CYCLES') / mb if mb else float('nan')):.2f} | "
                   f"{val('sq2', k, 'SQ_LDS_BANK_CONFLICT') / max(1.0, val('sq2', k, 'SQ_LDS_IDX_ACTIVE')):.3f} |")
    out += ["", "## Memory side", "",
            "`HBM GB/s` = (2 x FETCH_SIZE + WRITE_SIZE) / duration (FETCH_SIZE doubled per the gfx950 correction of "
            "/opt/skills/guides/MI355X_MICROARCH.md, HBM section; both in KiB); `L2->CU` = TCP_TCC_READ_REQ x 64 B / duration, the "
            "read requests the CUs' vector L1s sent to L2 (64-B requests); L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS).", "",
            "| kernel | us | read MB (2x FETCH) | write MB | HBM GB/s | TCP->TCC read req | L2->CU MB | L2->CU GB/s | per-CU B/clk (at 2.4 GHz, CUs used) | L2 hit |",
            "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for k in kernels:
        d = passes["tcc"][2].get(k) or passes["sq1"][2][k]
        us = d[0] / max(1, d[1])
        rd, wr = 2 * val("fetch", k, "FETCH_SIZE") * 1024, val("write", k, "WRITE_SIZE") * 1024
        req = val("tcc", k, "TCP_TCC_READ_REQ_sum")
        hit, miss = val("tcc", k, "TCC_HIT_sum"), val("tcc", k, "TCC_MISS_sum")
        waves = val("sq1", k, "SQ_WAVES")
        cus = 128 if k.startswith(("tail7", "proj_patch", "se_small")) else 256
        out.append(f"| `{k}` | {us:.1f} | {rd / 1e6:.1f} | {wr / 1e6:.1f} | {(rd + wr) / us / 1e3:.0f} | {req:.0f} | {req * 128 / 1e6:.1f} | "
                   f"{req * 128 / us / 1e3:.0f} | {req * 128 / (us * 1e-6 * 2.4e9) / cus:.1f} | {hit / max(1.0, hit + miss):.3f} |")
    dst = ROOT / "profiles" / f"{tag}_sq_counters.md"
    dst.write_text("\n".join(out) + "\n")
    print(dst)


if __name__ == "__main__":
    main()
