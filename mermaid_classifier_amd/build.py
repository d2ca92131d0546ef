"""Compile libmermaid_mi355.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m mermaid_classifier_amd.build [--force]

One object per translation unit (the kernels are five .hip files by layer group, plus the trainer and the C-ABI / schedule),
compiled in parallel and only when the source or one of its headers is newer than the object; then one link.  Objects live in
csrc/_obj/ (git-ignored and gpurun-ignored: only the linked library travels to the GPU box).
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OBJ = CSRC / "_obj"
OUT = HERE / "libmermaid_mi355.so"
KERNEL_HEADERS = ["device_common.h", "kernels.h"]
# translation unit -> headers it includes
SOURCES = {
    "k_generic.hip": KERNEL_HEADERS,
    "k_mbconv.hip": KERNEL_HEADERS,
    "k_early.hip": KERNEL_HEADERS,
    "k_mid.hip": KERNEL_HEADERS,
    "k_tail.hip": KERNEL_HEADERS,
    "trainer.hip": ["../../include/mmc.h"],
    "mmc_api.cpp": ["kernels.h", "../../include/mmc.h"],
    "mmc_dist.cpp": ["../../include/mmc.h"],
}
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC",
         "-mllvm", "-amdgpu-mfma-vgpr-form",   # MFMA results land in VGPRs: no v_accvgpr_read per element
         "-Wno-unused-result", "-Wno-unused-value", "-pthread"]


def _mtime(p: Path) -> float:
    return p.resolve().stat().st_mtime


def _stale(src: str) -> bool:
    obj = OBJ / (src + ".o")
    if not obj.is_file():
        return True
    t = obj.stat().st_mtime
    return any(_mtime(CSRC / d) > t for d in [src] + SOURCES[src])


def needs_build() -> bool:
    if not OUT.is_file():
        return True
    t = OUT.stat().st_mtime
    return any(_mtime(CSRC / d) > t for src, deps in SOURCES.items() for d in [src] + deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        raise RuntimeError("hipcc not found: cannot build libmermaid_mi355.so")
    OBJ.mkdir(exist_ok=True)

    def compile_one(src: str) -> None:
        cmd = [hipcc] + FLAGS + ["-c", "-o", str(OBJ / (src + ".o")), src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, cwd=str(CSRC), check=True)

    todo = [s for s in SOURCES if force or _stale(s)]
    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 1) or 1) as pool:
        list(pool.map(compile_one, todo))   # (re-raises the first failure)
    link = [hipcc, "--offload-arch=gfx950", "-fPIC", "-pthread", "-shared", "-o", str(OUT)] + [str(OBJ / (s + ".o")) for s in SOURCES]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, cwd=str(CSRC), check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
