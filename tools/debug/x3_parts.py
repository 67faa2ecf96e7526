#!/usr/bin/env python3
"""What bounds gemm3_dma_kernel: debug builds of the library with one part of the kernel removed (-DTVAE_X3DBG=1 no LDS-DMA
inside the k-loop, =2 no MFMAs, =3 no epilogue), timed on the forward shapes.  `python tools/debug/x3_parts.py build` in the
container, then `python tools/debug/x3_parts.py` on the GPU."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = lambda v: os.path.join(ROOT, "tools", "debug", "libtvae_x3dbg%d.so" % v)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    src = os.path.join(ROOT, "transvae_amd", "csrc")
    objs = [os.path.join(ROOT, "transvae_amd", "lib", f) for f in os.listdir(os.path.join(ROOT, "transvae_amd", "lib")) if f.endswith(".o") and f != "gemm3_dma.o"]
    for v in (1, 2, 3, 4, 5):      # 4, 5: four LDS stages (three k-tiles in flight), whole kernel / without its MFMAs
        defs = ["-DTVAE_X3DBG=%d" % v] if v < 4 else ["-DTVAE_X3_STAGES=4", "-DTVAE_X3DBG=%d" % (0 if v == 4 else 2)]
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-c", os.path.join(src, "gemm3_dma.hip"), "-o", "/tmp/gemm3_dma_dbg.o", "--offload-arch=gfx950", "-O3",
                               "-fPIC", "-std=c++17"] + defs)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-o", lib(v), "/tmp/gemm3_dma_dbg.o"] + objs + ["--offload-arch=gfx950", "-pthread"])
        print(lib(v))
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, ROOT)
    import torch
    import bench
    from transvae_amd import ops
    ops.set_compute_dtype("f32x3")
    M = 63500
    for N, K in ((1024, 256), (256, 1024), (256, 256)):
        x = torch.randn(M, K, device="cuda"); w = torch.nn.Parameter(torch.randn(N, K, device="cuda") * 0.05, requires_grad=False)
        b = torch.zeros(N, device="cuda"); y = ops.rows_buffer(M, N, "cuda")
        t = bench.time_launches(lambda: ops.linear_fwd(x, w, b, out=y), 20)
        print("  %-12s x[63500,%4d] @ W[%4d,%4d]^T + b  %7.1f us" % (os.environ.get("X3_TAG", ""), K, N, K, t * 1e6), flush=True)
    sys.exit(0)
for tag, path in (("whole kernel", None), ("no DMA", lib(1)), ("no MFMA", lib(2)), ("no epilogue", lib(3)), ("4 stages", lib(4)),
                  ("4 st no MFMA", lib(5))):
    env = dict(os.environ, X3_TAG=tag)
    if path:
        env["TVAE_LIB_PATH"] = path
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "one"], env=env)
