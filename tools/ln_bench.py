"""Time tvae_layernorm_fwd / _bwd on the train-step shape (63500 x 256) and report the HBM rate."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transvae_amd import _lib, ops  # noqa: E402


def main():
    rows, d = 63500, 256
    dev = torch.device("cuda:0")
    x, dy, res = (torch.randn(rows, d, device=dev) for _ in range(3))
    a, b = torch.rand(d, device=dev) + 0.5, torch.randn(d, device=dev)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    da, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    # rotate over several buffers so that the 256 MB last-level cache does not flatter the numbers
    xs = [torch.randn(rows, d, device=dev) for _ in range(8)]

    def timeit(fn, n=40):
        for i in range(5):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    s = ops._stream()
    t = timeit(lambda i: _lib.call("tvae_layernorm_fwd", ops._p(xs[i % 8]), ops._p(a), ops._p(b), ops._p(y), rows, d, 1e-6, s))
    print("ln_fwd  %.1f us  %.2f TB/s" % (t, 2 * rows * d * 4 / t / 1e6))
    t = timeit(lambda i: _lib.call("tvae_layernorm_bwd", ops._p(xs[i % 8]), ops._p(a), ops._p(xs[(i + 3) % 8]), None, ops._p(dx),
                                   ops._p(da), ops._p(db), rows, d, 1e-6, s))
    print("ln_bwd  %.1f us  %.2f TB/s (3 streams)" % (t, 3 * rows * d * 4 / t / 1e6))
    t = timeit(lambda i: _lib.call("tvae_layernorm_bwd", ops._p(xs[i % 8]), ops._p(a), ops._p(xs[(i + 3) % 8]), ops._p(xs[(i + 5) % 8]),
                                   ops._p(dx), ops._p(da), ops._p(db), rows, d, 1e-6, s))
    print("ln_bwd+res  %.1f us  %.2f TB/s (4 streams)" % (t, 4 * rows * d * 4 / t / 1e6))


if __name__ == "__main__":
    main()
