#!/usr/bin/env python3
"""Experiment: does the one-query decode attention (HBM-bound) run BESIDE the decode GEMMs (MFMA-bound) when both are
resident, and does a static CU partition (hipExtStreamCreateWithCUMask) make them?  Times, for n latents:
  attention alone / GEMMs alone / both on two plain streams / both on two streams with complementary CU masks.
python tools/debug/cu_mask_overlap.py [n]"""
import ctypes as C
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transvae_amd import ops
from transvae_amd._lib import call

hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
d, h, T, keys = 256, 4, 126, 63
torch.manual_seed(0)
qkv = torch.randn(n, 3 * d, device=dev)
ks = torch.randn(n, T, d, device=dev)
vs = torch.randn(n, T, d, device=dev)
att = torch.empty(n, d, device=dev)
x = torch.randn(n, d, device=dev)
w1 = torch.randn(1024, d, device=dev) * 0.05
b1 = torch.zeros(1024, device=dev)
w2 = torch.randn(d, 1024, device=dev) * 0.05
b2 = torch.zeros(d, device=dev)
hid = torch.empty(n, 1024, device=dev)
y = torch.empty(n, d, device=dev)
_p = lambda t: C.c_void_p(t.data_ptr())


def attn(reps):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(reps):
        call("tvae_decode_attn", _p(qkv), 3 * d, _p(ks), _p(vs), d, T, _p(att), d, None, 0, keys, n, h, d // h,
             1.0 / math.sqrt(d // h), st)


def gemms(reps):
    for _ in range(reps):
        ops.linear_fwd(x, w1, b1, relu=True, out=hid)
        ops.linear_fwd(hid, w2, b2, out=y)


def masked_stream(pred):
    words = (C.c_uint32 * 8)()
    cnt = 0
    for i in range(256):
        if pred(i):
            words[i // 32] |= 1 << (i % 32)
            cnt += 1
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value), cnt


def wall(fa, fb, sa, sb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa:
        with torch.cuda.stream(sa):
            fa()
    if fb:
        with torch.cuda.stream(sb):
            fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


RA, RG = 24, 48
with torch.no_grad():
    for _ in range(3):
        attn(RA); gemms(RG)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    ta = min(wall(lambda: attn(RA), None, s1, s2) for _ in range(3))
    tg = min(wall(None, lambda: gemms(RG), s1, s2) for _ in range(3))
    tb = min(wall(lambda: attn(RA), lambda: gemms(RG), s1, s2) for _ in range(3))
    tb2 = min(wall(lambda: gemms(RG), lambda: attn(RA), s1, s2) for _ in range(3))
    print("n=%d  attention alone %.2f ms (%d launches)  GEMMs alone %.2f ms (%d pairs)  sum %.2f" % (n, ta, RA, tg, RG, ta + tg))
    print("two plain streams: %.2f ms (attention issued first), %.2f ms (GEMMs issued first)" % (tb, tb2))
    # i = 32 q + 8 a + r: patterns that give every XCD the same share whether the bit -> XCD map is i % 8 or i / 32
    for num in (2, 3, 4, 5):
        in_a = lambda i, num=num: ((i % 8) + (i // 32)) % 8 < num
        sa, ca = masked_stream(in_a)
        sg, cg = masked_stream(lambda i: not in_a(i))
        for _ in range(2):
            wall(lambda: attn(RA), lambda: gemms(RG), sa, sg)
        ma = min(wall(lambda: attn(RA), None, sa, sg) for _ in range(3))
        mg = min(wall(None, lambda: gemms(RG), sa, sg) for _ in range(3))
        mb = min(wall(lambda: attn(RA), lambda: gemms(RG), sa, sg) for _ in range(3))
        print("masks %3d | %3d CUs: attention alone %.2f ms, GEMMs alone %.2f ms, both %.2f ms" % (ca, cg, ma, mg, mb), flush=True)
