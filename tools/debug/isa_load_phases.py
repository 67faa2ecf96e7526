#!/usr/bin/env python3
"""How many memory round trips does a kernel's straight path chain?  Per kernel of an object file: the number of
load -> s_waitcnt vmcnt -> load alternations in program order (loops counted once).  One phase is the ideal for a
latency-sized kernel; each extra one is a round trip the loads before it could have shared.
python tools/debug/isa_load_phases.py transvae_amd/lib/rowops.o"""
import re, subprocess, sys, os, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
obj = os.path.abspath(sys.argv[1])
with tempfile.TemporaryDirectory() as d:
    t = os.path.join(d, "x.o"); subprocess.check_call(["cp", obj, t])
    subprocess.check_call([LLVM + "/llvm-objdump", "--offloading", t], stdout=subprocess.DEVNULL, cwd=d)
    co = [f for f in os.listdir(d) if "gfx950" in f][0]
    asm = subprocess.check_output([LLVM + "/llvm-objdump", "-d", os.path.join(d, co)], text=True)
out, name = [], None
for line in asm.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
    if m:
        if name: out.append((phases, loads, name))
        name, phases, loads, state = m.group(1), 0, 0, "idle"
        continue
    if name is None: continue
    if "global_load" in line or "buffer_load" in line:
        loads += 1
        if state != "loading": phases += 1
        state = "loading"
    elif "s_waitcnt" in line and "vmcnt" in line and state == "loading":
        state = "waited"
if name: out.append((phases, loads, name))
for phases, loads, name in sorted(out, reverse=True):
    print("%3d load phases  %4d loads  %s" % (phases, loads, subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:140]))
