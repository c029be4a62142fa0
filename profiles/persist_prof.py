#!/usr/bin/env python3
"""Reads a ZPAQ_AMD_PERSIST_PROF file (engine.cpp launch_pipe_persist): per unit wavefront of the persistent encoder launch the
time it spent waiting for other units and the time it spent working, 100 MHz ticks.  Prints per unit type and role the mean /
max over the groups, and per workgroup flavour the SIMD sums."""
import struct, sys, collections
import numpy as np
b = open(sys.argv[1], "rb").read()
ng, nslot, waves, wpg = struct.unpack_from("<4Q", b, 0)
a = np.frombuffer(b, np.uint64, offset=32).reshape(ng, nslot, 4)
names = {0: "hcomp", 1: "row", 2: "light", 3: "icm", 4: "isse", 5: "mix"}
print(f"groups {ng}, slots per group {nslot} ({wpg} workgroups x {waves} wavefronts)")
print("slot wg wave simd  unit           work ms (mean / max)   wait ms (mean)   chunks")
for s in range(nslot):
    w = a[:, s, :]
    if not w[:, 2].any():
        continue
    tag = int(w[0, 3])
    kind, role, unit = tag >> 32, (tag >> 16) & 0xFFFF, tag & 0xFFFF
    print(f"{s:4d} {s // waves:2d} {s % waves:4d} {s % waves % 4:4d}  {names.get(kind, '?'):5s} role {role:2d} u{unit:3d}  "
          f"{w[:, 1].mean() / 1e5:8.1f} / {w[:, 1].max() / 1e5:8.1f}   {w[:, 0].mean() / 1e5:8.1f}   {int(w[0, 2])}")
print("per workgroup flavour and SIMD: sum of the mean work of its wavefronts (ms)")
for f in range(wpg):
    sums = [0.0] * 4
    for wv in range(waves):
        s = f * waves + wv
        sums[wv % 4] += a[:, s, 1].mean() / 1e5
    print(f"  flavour {f}: " + "  ".join(f"{x:8.1f}" for x in sums))
