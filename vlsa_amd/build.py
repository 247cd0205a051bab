"""Build libvlsa_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by
``python -m vlsa_amd.build``.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libvlsa_hip.so")
SOURCES = ["vlfan_partial.hip", "vlfan_partial_dma.hip", "vlfan_batch.hip", "vlfan_batch_f32.hip", "vlfan_backward.hip", "vlfan_backward_batch.hip", "vlfan_backward_batch_f32.hip", "vlfan_tail.hip", "mil_pool.hip", "ingest.hip", "surv_loss.hip", "gated_scores.hip", "feat_proj.hip", "text_tower.hip", "prompt_sentences.hip"]
HEADERS = ["vlsa_common.h", "vlfan_mfma_common.h", os.path.join("..", "..", "include", "vlsa_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 into one shared library. Returns its path."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wall", "-Wno-unused-function",
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libvlsa_hip.so")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
