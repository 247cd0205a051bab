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
SOURCES = ["vlfan_partial.hip", "vlfan_partial_dma.hip", "vlfan_batch.hip", "vlfan_batch_f32.hip", "vlfan_backward.hip", "vlfan_backward_batch.hip", "vlfan_backward_batch_f32.hip", "vlfan_tail.hip", "mil_pool.hip", "ingest.hip", "surv_loss.hip", "adam.hip", "gated_scores.hip", "gated_scores_tile.hip", "feat_proj.hip", "text_tower.hip", "prompt_sentences.hip", "mlp_backward.hip", "vlfan_dx.hip", "xchg.hip"]
HEADERS = ["vlsa_common.h", "vlfan_mfma_common.h", "gated_scores.h", os.path.join("..", "..", "include", "vlsa_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
STAMP = os.path.join(LIB_DIR, "flags.stamp")


def _flags():
    """compile flags; VLSA_EXTRA_HIPCC_FLAGS adds debug switches (e.g. -DVLSA_TT_DEBUG: cycle stamps in the text GEMM, whose
    results are "wrong: timing only")"""
    return BASE_FLAGS + os.environ.get("VLSA_EXTRA_HIPCC_FLAGS", "").split()


def _flag_tag() -> str:
    import hashlib
    return hashlib.sha1(" ".join(_flags()).encode()).hexdigest()[:10]


def _obj_dir() -> str:
    """objects of a non-default flag set live in their own directory: a debug build never leaves objects a later normal build
    would link (and the other way round)"""
    extra = os.environ.get("VLSA_EXTRA_HIPCC_FLAGS", "").split()
    return os.path.join(LIB_DIR, "obj" if not extra else "obj-" + _flag_tag())


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _stale() -> bool:
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if _newer(LIB_PATH, deps):
        return True
    try:        # the library on disk was linked from another flag set (debug build <-> normal build)
        return open(STAMP).read().strip() != _flag_tag()
    except OSError:
        return True


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 (one object per source, stale ones only, in parallel) and link them into one shared
    library.  Returns its path."""
    if not force and not _stale():
        return LIB_PATH
    OBJ_DIR = _obj_dir()
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    common = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    flags = _flags()
    jobs = []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
        if force or _newer(obj, [os.path.join(CSRC, src)] + common):
            cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append((src, cmd))
    width = max(1, min(len(jobs), os.cpu_count() or 1))
    for i in range(0, len(jobs), width):
        procs = [(src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)) for src, cmd in jobs[i:i + width]]
        for src, pr in procs:
            out, _ = pr.communicate()
            if pr.returncode != 0:
                sys.stderr.write(out)
                raise RuntimeError(f"hipcc failed on {src}")
            if verbose and out.strip():
                print(out)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in SOURCES],
            "-o", LIB_PATH + ".tmp"]
    res = subprocess.run(link, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed linking libvlsa_hip.so")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(STAMP, "w") as f:
        f.write(_flag_tag() + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv, verbose=True))
