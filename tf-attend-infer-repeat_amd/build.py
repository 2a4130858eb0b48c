"""Builds libair_hip.so (gfx950) in-tree: python tf-attend-infer-repeat_amd/build.py [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off: every fp32 op rounds once
in source order (the sampler / loss kernels mirror the reference's op order)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))
HDR = sorted(glob.glob(os.path.join(HERE, "csrc", "*.h"))) + [os.path.join(ROOT, "include", "air_hip.h")]
OUT = os.path.join(HERE, "libair_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in SRC + HDR)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + SRC + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", OUT)
