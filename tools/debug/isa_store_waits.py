#!/usr/bin/env python3
"""Which kernels wait for their own stores?  For every kernel of an object file: the number of s_waitcnt vmcnt(0) that sit
between two global stores / atomics with no global load in between -- each one is a full round trip to memory spent waiting
for an acknowledgement nobody needs (the compiler's count of outstanding operations, merged conservatively where branches
join).  python tools/debug/isa_store_waits.py transvae_amd/lib/attention.o"""
import re, subprocess, sys, os, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.abspath(sys.argv[1])
with tempfile.TemporaryDirectory() as d:
    t = os.path.join(d, "x.o"); subprocess.check_call(["cp", obj, t])
    subprocess.check_call([LLVM + "/llvm-objdump", "--offloading", t], stdout=subprocess.DEVNULL, cwd=d)
    co = [f for f in os.listdir(d) if "gfx950" in f][0]
    asm = subprocess.check_output([LLVM + "/llvm-objdump", "-d", os.path.join(d, co)], text=True)
name, state, sus, stores = None, None, 0, 0
out = []
for line in asm.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
    if m:
        if name: out.append((sus, stores, name))
        name, state, sus, stores = m.group(1), None, 0, 0
        continue
    if "global_store" in line or "global_atomic" in line or "buffer_store" in line:
        stores += 1
        state = "store"
    elif "global_load" in line or "buffer_load" in line:
        state = "load"
    elif "s_waitcnt" in line and "vmcnt(0)" in line and state == "store":
        sus += 1
        state = "waited"
if name: out.append((sus, stores, name))
for sus, stores, name in sorted(out, reverse=True):
    if sus: print("%4d waits after stores  %4d stores  %s" % (sus, stores, subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:150]))
