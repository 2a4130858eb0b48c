"""Builds libair_hip.so (gfx950) in-tree: python tf-attend-infer-repeat_amd/build.py [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off: every fp32 op rounds once
in source order (the sampler / loss kernels mirror the reference's op order).
Every csrc/*.hip is compiled to its own object (in parallel, re-compiled only when
it or a header is newer than the object), then linked."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
HDR = sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) + [os.path.join(ROOT, "include", "air_hip.h")]
OUT = os.path.join(HERE, "libair_hip.so")
OBJ = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]


def _obj(src):
    return os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(p) > t for p in deps)


def needs_build():
    return _stale(OUT, SRC + HDR)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in SRC if force or _stale(_obj(s), [s] + HDR)]

    def compile_one(src):
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in SRC] + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
