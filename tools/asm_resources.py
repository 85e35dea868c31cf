#!/usr/bin/env python
"""Per-function register / scratch use of the gfx950 code object: parse the `-save-temps` assembly of mwgpu.hip.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -amdgpu-mfma-vgpr-form -save-temps -o /tmp/x.so metaworld_amd/csrc/mwgpu.hip
       python tools/asm_resources.py mwgpu-hip-amdgcn-amd-amdhsa-gfx950.s [rows=50]
       python tools/asm_resources.py mwgpu-hip-amdgcn-amd-amdhsa-gfx950.s --calls
--calls: every NON-INLINED callee of the lane programs (targets of s_swappc_b64) with its number of call sites, checked against the
list below of how each is entered.  A non-inlined function that uses all 256 VGPRs must be entered by every live lane of the wave or
by none (DESIGN.md 5 "register hazard"); a new callee that is not on the list fails the check (exit status 1) until someone has
looked at its call sites."""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
# how every non-inlined callee is entered: "wave" = by every live lane of the wave or by none (wave-uniform control flow around the call;
# ghost lanes keep partial workgroups full), "masked" = under a partial EXEC mask by design, covered by the sub-lane canary (flag 8)
ENTERED = {"forward": "wave", "substep": "wave", "forward_dynamics": "wave", "lane_step": "wave", "kinematics": "wave", "crb": "wave",
           "smooth_forces": "wave", "collision": "wave", "collide_pair": "wave", "make_constraints": "wave", "solve": "wave", "task_evaluate": "wave",
           "solve_wave": "wave", "scripted_policy": "wave", "integrate": "wave", "integrate_full": "wave", "solve_env": "wave", "collision_gather": "wave",
           "update_constraint": "masked", "newton_direction": "masked"}
if len(sys.argv) > 2 and sys.argv[2] == "--calls":
    syms = re.findall(r"s_add_u32 s\d+, s\d+, (_Z\w+)@rel32@lo", s)
    names = subprocess.run(["c++filt"] + syms, capture_output=True, text=True).stdout.split("\n")
    count = {}
    for n in names:
        if n:
            base = re.sub(r"<.*", "", n.split("(")[0].split("::")[-1]).strip()
            base = re.sub(r"^.* ", "", base)
            count[base] = count.get(base, 0) + 1
    bad = 0
    for k, v in sorted(count.items(), key=lambda kv: -kv[1]):
        how = ENTERED.get(k, "NOT ON THE LIST")
        bad += k not in ENTERED
        print(f"{v:4d} call sites  {k:28s} {how}")
    sys.exit(1 if bad else 0)
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
