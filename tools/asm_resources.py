#!/usr/bin/env python
"""Per-function register / scratch use of the gfx950 code object: parse the `-save-temps` assembly of mwgpu.hip.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -save-temps -o /tmp/x.so metaworld_amd/csrc/mwgpu.hip
       python tools/asm_resources.py mwgpu-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
rows = []
for m in re.finditer(r"; -- End function\n(.*?); Function info:\n(.*?)\n\t\.text", s, flags=re.S):
    sets, info = m.group(1), m.group(2)
    nm = re.search(r"\.set \.?L?(\S+?)\.num_vgpr", sets)
    if not nm:
        continue
    g = lambda k: int(re.search(k + r"[:=]\s*(\d+)", info).group(1))
    rows.append((g("codeLenInByte "), g("NumVgprs"), g("NumAgprs"), g("ScratchSize"), nm.group(1)))
names = subprocess.run(["c++filt"] + [r[4] for r in rows], capture_output=True, text=True).stdout.split("\n")
out = []
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "")
    out.append(r[:4] + (n[:140],))
print(f"{'bytes':>8s} {'vgpr':>5s} {'agpr':>5s} {'scratch':>8s}  function")
for o in sorted(out, reverse=True)[: int(sys.argv[2]) if len(sys.argv) > 2 else 50]:
    print(f"{o[0]:8d} {o[1]:5d} {o[2]:5d} {o[3]:8d}  {o[4]}")
