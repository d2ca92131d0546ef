#!/usr/bin/env python3
"""Condense rocprofv3 output merged under gpurun_out/ into the tracked summaries under profiles/.

  python tools/summarize_profiles.py <round-tag> <kernel-stats-dir> [<pmc-fetch-dir> <pmc-write-dir>]
"""
import collections
import csv
import glob
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
    return name.replace("void ", "")[:60]


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    import os
    latest = lambda pat: max(glob.glob(pat), key=os.path.getmtime)
    rows = list(csv.DictReader(open(latest(f"{stats_dir}/*/*kernel_stats.csv"))))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    cmd = os.environ.get("SUMMARY_CMD", "python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-passes 1 --spread-blocks 1` "
                         "(passes of 256 patches, 2 lanes of 128: every launch covers 128 patches; both lanes running")
    out = [f"# rocprofv3 --kernel-trace --stats summary ({tag})", "",
           "command: `rocprofv3 --kernel-trace --stats --output-format csv -- " + cmd + ")", "",
           f"total kernel time {total / 1e6:.2f} ms", "",
           "| kernel | calls | avg us | total ms | % |", "|---|---:|---:|---:|---:|"]
    for r in rows:
        if float(r["Percentage"]) < 0.05:
            continue
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | "
                   f"{float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['Percentage']):.1f} |")
    if len(sys.argv) >= 5:
        def load(d, counter):
            acc = collections.defaultdict(lambda: [0.0, 0])
            for r in csv.DictReader(open(latest(f"{d}/*/*counter_collection.csv"))):
                if r["Counter_Name"] == counter:
                    k = short(r["Kernel_Name"])
                    acc[k][0] += float(r["Counter_Value"])
                    acc[k][1] += 1
            return acc
        f, w = load(sys.argv[3], "FETCH_SIZE"), load(sys.argv[4], "WRITE_SIZE")
        out += ["", "## HBM traffic per launch (PMC, separate passes: `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`)", "",
                "FETCH_SIZE / WRITE_SIZE are in KiB.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE on "
                "gfx950 counts exactly half the bytes of wide (16 B/lane) coalesced reads, so `read MB` = 2 x FETCH_SIZE; "
                "WRITE_SIZE is exact for 16-B stores.", "",
                "| kernel | launches | read MB/launch (2x FETCH) | write MB/launch |", "|---|---:|---:|---:|"]
        for k in sorted(f, key=lambda k: -f[k][0]):
            if f[k][0] / 1024 < 1 and w[k][0] / 1024 < 1:
                continue
            out.append(f"| `{k}` | {f[k][1]} | {2 * f[k][0] / f[k][1] / 1024:.1f} | {w[k][0] / max(1, w[k][1]) / 1024:.1f} |")
    if len(sys.argv) >= 5:
        import json
        traffic = {k: {"read_bytes_per_launch": 2 * f[k][0] / f[k][1] * 1024, "write_bytes_per_launch": w[k][0] / max(1, w[k][1]) * 1024,
                       "launches": f[k][1]} for k in f}
        (ROOT / "profiles" / f"{tag}_pmc_traffic.json").write_text(json.dumps(traffic, indent=1))
    dst = ROOT / "profiles" / f"{tag}_rocprof_summary.md"
    dst.write_text("\n".join(out) + "\n")
    print(dst)


if __name__ == "__main__":
    main()
