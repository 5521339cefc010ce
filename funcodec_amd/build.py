"""Build the gfx950 shared library in-tree (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libfuncodec_amd.so")
SOURCES = ["kernels.hip", "engine.hip"]
HEADERS = ["kernels.h", os.path.join("..", "..", "include", "funcodec_amd.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    timeline = bool(os.environ.get("FC_TIMELINE"))
    # profiling build (phase timestamps inside the conv kernel, fc_debug_timeline): a SEPARATE file, selected with FC_LIB=<path>
    out = LIB_PATH.replace(".so", "_timeline.so") if timeline else LIB_PATH
    if not force and not timeline and not needs_build():
        return LIB_PATH
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    if timeline:
        cmd.insert(1, "-DFC_TIMELINE")
    for flag in os.environ.get("FC_BUILD_DEFINES", "").split():      # tuning aid: experimental -D switches into a separate file
        cmd.insert(1, "-D" + flag)
    if os.environ.get("FC_BUILD_OUT"):
        out = os.environ["FC_BUILD_OUT"]
        cmd[cmd.index("-o") + 1] = out
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
