#!/usr/bin/env python3
"""Fit of the packing's cost model (host/codegen.cpp plan_persistent) from per-unit profiles of the persistent encoder launch
(ZPAQ_AMD_PERSIST_PROF files of the -m5 headline under several packings): a workgroup = a compute unit, its time = the busy
time of its slowest wavefront, explained by WHAT the workgroup holds -- non-negative least squares of
    T(workgroup) = alpha + sum over unit types (weight(type) x wavefronts of that type in the workgroup).
    python profiles/r06/fit_packing.py gpurun_out/r06/c7_prof_*.bin"""
import struct, sys
import numpy as np
from scipy.optimize import nnls

LIGHT = {0: "CM(lds)", 1: "MATCH", 2: "MIX2(lds)", 3: "SSEbits", 4: "SSEbits", 5: "SSEbits", 6: "SSEbits", 7: "MIX2(1)", 8: "CODER"}
ONDIE_ROWS = {2, 3}          # -m5: ROW units of components 3 (2 KiB table) and 4 (128 KiB)


def unit_type(kind, role):
    if kind == 0: return "HCOMP"
    if kind == 1: return "ROW on-die" if role in ONDIE_ROWS else "ROW hbm"
    if kind == 2: return LIGHT.get(role, "light")
    if kind == 3: return "ICM map"
    if kind == 4: return "ISSE map"
    return "MIX8 half" if role == 0 else "MIX16 half"


def main():
    rows, ts, names = [], [], []
    types = ["ROW hbm", "ROW on-die", "MIX8 half", "MIX16 half", "SSEbits", "MATCH", "ISSE map", "ICM map", "HCOMP", "CM(lds)", "MIX2(lds)", "MIX2(1)", "CODER"]
    for path in sys.argv[1:]:
        b = open(path, "rb").read()
        ng, nslot, waves, wpg = struct.unpack_from("<4Q", b, 0)
        a = np.frombuffer(b, np.uint64, offset=32).reshape(ng, nslot, 4)
        launch = float((a[:, :, 0] + a[:, :, 1]).max()) / 1e5
        for f in range(wpg):
            cnt = dict.fromkeys(types, 0)
            worst = 0.0
            for w in range(waves):
                s = f * waves + w
                if not a[:, s, 2].any():
                    continue
                tag = int(a[0, s, 3])
                cnt[unit_type(tag >> 32, (tag >> 16) & 0xFFFF)] += 1
                worst = max(worst, float(a[:, s, 1].mean()) / 1e5)
            rows.append([cnt[t] for t in types]); ts.append(worst); names.append((path.split("/")[-1], f, launch))
    A = np.array(rows, float)
    y = np.array(ts)
    A1 = np.hstack([np.ones((len(y), 1)), A])
    x, res = nnls(A1, y)
    pred = A1 @ x
    print(f"{len(y)} workgroups of {len(sys.argv) - 1} launches; alpha = {x[0]:.0f} ms; rms residual {np.sqrt(((pred - y) ** 2).mean()):.0f} ms of {y.mean():.0f} ms mean")
    for t, w in zip(types, x[1:]):
        print(f"  {t:12s} {w:7.1f} ms per wavefront   (relative to a ROW on HBM: {w / max(x[1], 1e-9):.2f})")
    print("per launch: slowest workgroup measured / predicted, launch time")
    seen = {}
    for (p, f, launch), yy, pp in zip(names, y, pred):
        d = seen.setdefault(p, [0, 0, launch])
        d[0] = max(d[0], yy); d[1] = max(d[1], pp)
    for p, (m, pr, launch) in seen.items():
        print(f"  {p:34s} {m:7.0f} {pr:7.0f} {launch:7.0f}")


if __name__ == "__main__":
    main()
