"""Build libtvae_hip.so (gfx950) in-tree with hipcc.  No GPU needed: hipcc cross-compiles."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libtvae_hip.so")
SOURCES = ["gemm.hip", "gemm3.hip", "gemm3_dma.hip", "gemm_dma.hip", "gemm_wide.hip", "rowops.hip", "attention.hip", "lossopt.hip", "dataprep.hip", "gru.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_common.h"),
               os.path.join(HERE, "..", "include", "tvae_kernels.h")]
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + headers):
            jobs.append([_hipcc(), "-c", sp, "-o", op] + FLAGS)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for err in ex.map(run, jobs):
            if verbose and err.strip():
                print(err)
    if jobs or force or _stale(LIB, objs):
        run([_hipcc(), "-shared", "-o", LIB] + objs + ["--offload-arch=gfx950", "-pthread"])
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
