"""Compile libmermaid_mi355.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m mermaid_classifier_amd.build [--force]
"""

from __future__ import annotations

import os
import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "libmermaid_mi355.so"
SOURCES = ["kernels.hip", "trainer.hip", "mmc_api.cpp"]
DEPS = SOURCES + ["kernels.h", "../../include/mmc.h"]


def needs_build() -> bool:
    if not OUT.is_file():
        return True
    t = OUT.stat().st_mtime
    return any((CSRC / d).resolve().stat().st_mtime > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not needs_build():
        return OUT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        raise RuntimeError("hipcc not found: cannot build libmermaid_mi355.so")
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-mllvm", "-amdgpu-mfma-vgpr-form",   # MFMA results land in VGPRs: no v_accvgpr_read per element
           "-Wno-unused-result", "-Wno-unused-value", "-pthread", "-o", str(OUT)] + SOURCES
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=str(CSRC), check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
