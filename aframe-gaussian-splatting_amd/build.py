"""Build the product: csrc/*.hip + csrc/*.cpp -> csrc/libgs_splat_hip.so (gfx950 code objects, C ABI of
include/gs_splat.h) and, when node headers are present, js/gs_splat_napi.node (the N-API addon).

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build container; the .so files are
git-ignored but travel to the GPU box with the working tree.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
JS = os.path.join(HERE, "js")
LIB = os.path.join(CSRC, "libgs_splat_hip.so")
ADDON = os.path.join(JS, "gs_splat_napi.node")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: the sort key is un-fused IEEE f64 and the projection has a fixed fp32 operation order
#   (gs_device_math.h); the kernels call fmaf() explicitly where fusion is wanted.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
             "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def build_lib(force=False, verbose=False):
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(ROOT, "include", "gs_splat.h")]
    if not force and not _stale(LIB, deps):
        return LIB
    objs = []
    for s in srcs:
        o = os.path.splitext(s)[0] + ".o"
        if force or _stale(o, deps):
            cmd = [HIPCC] + HIP_FLAGS + ["-x", "hip", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def node_include():
    for d in ("/usr/include/node", "/usr/local/include/node"):
        if os.path.exists(os.path.join(d, "node_api.h")):
            return d
    return None


def build_addon(force=False, verbose=False):
    """Raw-C N-API addon (no node-gyp): links against libgs_splat_hip.so with an $ORIGIN-relative rpath."""
    inc = node_include()
    src = os.path.join(JS, "gs_splat_napi.c")
    if inc is None or not os.path.exists(src):
        return None
    if not force and not _stale(ADDON, [src, LIB, os.path.join(ROOT, "include", "gs_splat.h")]):
        return ADDON
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall", "-I", inc, "-I", os.path.join(ROOT, "include"), "-o", ADDON, src,
           "-L", CSRC, "-lgs_splat_hip", "-Wl,-rpath,$ORIGIN/../csrc"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return ADDON


def build_all(force=False, verbose=False):
    lib = build_lib(force, verbose)
    addon = build_addon(force, verbose)
    return lib, addon


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
