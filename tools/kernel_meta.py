#!/usr/bin/env python3
"""Kernel metadata (VGPR count, spilled registers, private segment, LDS) of the gfx950 code objects embedded in a hipcc host object:

    python tools/kernel_meta.py diffphore_amd/csrc/dp_conv_node_k2.o

Finds the ELF images inside the object's .hip_fatbin section and prints `llvm-readelf --notes` fields per kernel (DESIGN.md 5.3 quotes them)."""
import re, subprocess, sys
# find embedded ELF code objects inside the host object's .hip_fatbin section and print kernel metadata
data = open(sys.argv[1], "rb").read()
idx = [m.start() for m in re.finditer(b"\x7fELF", data)]
n = 0
for i in idx[1:]:
    blob = data[i:]
    open("/tmp/_co.elf", "wb").write(blob)
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", "/tmp/_co.elf"], capture_output=True, text=True)
    if ".vgpr_count" not in r.stdout:
        continue
    cur = {}
    for line in r.stdout.splitlines():
        m = re.match(r"\s*[-]?\s*\.(name|vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|agpr_count|group_segment_fixed_size):\s*(.*)", line)
        if m:
            k, v = m.group(1), m.group(2).strip()
            if k == "name" and "name" in cur and "vgpr_count" in cur:
                print(cur); cur = {}
            if k == "name" and not v.startswith("_Z"):
                continue
            cur[k] = v
    if cur: print(cur)
