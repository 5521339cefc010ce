"""Build the gfx950 shared library in-tree (hipcc cross-compiles without a GPU).

The conv kernel template is instantiated in one translation unit per tile shape (csrc/conv_tile_*.hip); all
translation units compile in parallel and are linked into ONE library, funcodec_amd/libfuncodec_amd.so.
"""
from __future__ import annotations

import concurrent.futures
import glob
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libfuncodec_amd.so")
OBJ_DIR = os.path.join(CSRC, "_obj")
HEADERS = ["kernels.h", "conv_kernel.h", "laura_kernels.h", os.path.join("..", "..", "include", "funcodec_amd.h")]


def sources():
    # longest translation units first (the pool starts them in this order): the quad-layout tile units, then the rest
    tiles = sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "conv_tile*_*.hip")))
    heavy = [t for t in tiles if t.startswith("conv_tileq_") and "_m" not in t] + [t for t in tiles if t.endswith("_m01.hip")]
    rest = [t for t in tiles if t not in heavy]
    return heavy + ["kernels.hip", "laura_persist.hip", "laura_kernels.hip", "freq_kernels.hip", "engine.hip", "laura.hip"] + rest


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in sources() + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    timeline = bool(os.environ.get("FC_TIMELINE"))
    # profiling build (phase timestamps inside the conv kernel, fc_debug_timeline): a SEPARATE file, selected with FC_LIB=<path>
    out = LIB_PATH.replace(".so", "_timeline.so") if timeline else LIB_PATH
    defines = ["-DFC_TIMELINE", "-DFC_AB_KNOBS"] if timeline else []     # the timeline build is a tuning build: its A / B switches are live
    defines += ["-D" + f for f in os.environ.get("FC_BUILD_DEFINES", "").split()]      # tuning aid: experimental -D switches
    if os.environ.get("FC_BUILD_OUT"):
        out = os.environ["FC_BUILD_OUT"]
    if not force and not defines and out == LIB_PATH and not needs_build():
        return LIB_PATH
    # objects are cached per (output name, compile flags): a build with other -D switches must never reuse these objects
    import hashlib
    hipcc = _hipcc()
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + defines
    tag = "".join(c if c.isalnum() else "_" for c in os.path.basename(out)) + "_" + hashlib.sha1(" ".join(base[1:]).encode()).hexdigest()[:10]
    obj_dir = os.path.join(OBJ_DIR, tag)
    os.makedirs(obj_dir, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))

    def compile_one(src: str) -> str:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time):
            return obj
        cmd = base + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    srcs = sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, srcs))
    # link next to the target and rename over it: a process dlopen-ing the library concurrently sees the old or the new file, never
    # a half-written one
    tmp_out = f"{out}.{os.getpid()}.tmp"
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_out] + objs
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.run(link, check=True)
    os.replace(tmp_out, out)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
