#!/usr/bin/env python3
"""Where and when every workgroup of the pipelined encoder ran.

  ZPAQ_AMD_SPEC_DEFS=-DZPQ_TRACE ZPAQ_AMD_PIPE_TRACE=/tmp/trace.bin python bench.py --blocks 1024 --block-bytes 65536 \\
      --cpu-seconds 0 --api-blocks 0 --verify-blocks 0 --warmup 0 --steps 1
  python profiles/pipe_trace.py /tmp/trace.bin

The engine (device/engine.cpp::launch_pipe) writes four 64-bit words per workgroup and launch:
  [kernel << 56 | step << 32 | workgroup]  [XCC_ID << 32 | HW_ID]  [entry]  [exit]      (100 MHz reference clock)
HW_ID (gfx9): wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13].  Printed: per kernel the launches, the average /
maximum workgroup duration and the average kernel duration per step (first entry to last exit); how many wavefronts share
a SIMD (sampled at the middle of every workgroup's life); and how a workgroup's duration grows with the company it has."""
import sys

import numpy as np

NAMES = ["hcomp", "rows", "light", "icm", "isse", "mix"]


def load(path):
    a = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)
    a = a[(a[:, 2] != 0) & (a[:, 3] >= a[:, 2])]          # workgroups that never ran leave zeros
    rec = {
        "kernel": (a[:, 0] >> np.uint64(56)).astype(np.int64),
        "step": ((a[:, 0] >> np.uint64(32)) & np.uint64(0xFFFFFF)).astype(np.int64),
        "wg": (a[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64),
        "t0": a[:, 2].astype(np.int64), "t1": a[:, 3].astype(np.int64),
    }
    hw = (a[:, 1] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((a[:, 1] >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    rec["cu_key"] = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    rec["simd_key"] = rec["cu_key"] * 4 + simd
    return rec


def report(rec, out=sys.stdout):
    t_origin = rec["t0"].min()
    dur = (rec["t1"] - rec["t0"]) * 0.01                    # us
    print(f"{len(dur)} workgroup records, {len(np.unique(rec['simd_key']))} distinct SIMDs on {len(np.unique(rec['cu_key']))} CUs, "
          f"span {(rec['t1'].max() - t_origin) * 1e-5:.1f} ms", file=out)
    print(f"{'kernel':<7}{'launches':>9}{'wg avg us':>11}{'wg max us':>11}{'kernel avg us':>15}", file=out)
    for k, name in enumerate(NAMES):
        m = rec["kernel"] == k
        if not m.any():
            continue
        steps = rec["step"][m]
        order = np.argsort(steps, kind="stable")
        s_sorted = steps[order]
        first = np.r_[0, np.flatnonzero(np.diff(s_sorted)) + 1]
        k0 = np.minimum.reduceat(rec["t0"][m][order], first)
        k1 = np.maximum.reduceat(rec["t1"][m][order], first)
        print(f"{name:<7}{len(first):>9}{dur[m].mean():>11.1f}{dur[m].max():>11.1f}{((k1 - k0) * 0.01).mean():>15.1f}", file=out)
    # company on the SIMD: for every workgroup, the wavefronts resident on its SIMD at the middle of its life
    mid = (rec["t0"] + rec["t1"]) // 2
    company = np.zeros(len(mid), np.int64)
    order = np.argsort(rec["simd_key"], kind="stable")
    keys = rec["simd_key"][order]
    bounds = np.r_[0, np.flatnonzero(np.diff(keys)) + 1, len(keys)]
    for b0, b1 in zip(bounds[:-1], bounds[1:]):
        idx = order[b0:b1]
        t0, t1, mm = rec["t0"][idx], rec["t1"][idx], mid[idx]
        if len(idx) > 4000:                                  # long traces: a sample is enough
            pick = np.random.default_rng(0).choice(len(idx), 4000, replace=False)
        else:
            pick = np.arange(len(idx))
        company[idx[pick]] = ((t0[None, :] <= mm[pick, None]) & (t1[None, :] > mm[pick, None])).sum(1)
    seen = company > 0
    hist = np.bincount(company[seen])
    print("wavefronts sharing a SIMD (at the middle of a workgroup's life): " +
          "  ".join(f"{n}:{100.0 * c / seen.sum():.1f}%" for n, c in enumerate(hist) if n and c), file=out)
    print(f"{'kernel':<7}" + "".join(f"{'alone' if n == 1 else 'with %d' % (n - 1):>10}" for n in range(1, 7)) + "   (average workgroup us)", file=out)
    for k, name in enumerate(NAMES):
        m = (rec["kernel"] == k) & seen
        if not m.any():
            continue
        row = []
        for n in range(1, 7):
            mm = m & (company == n)
            row.append(f"{dur[mm].mean():>10.1f}" if mm.any() else f"{'-':>10}")
        print(f"{name:<7}" + "".join(row), file=out)


def selftest():
    """synthetic trace: two SIMDs, one crowded"""
    r = []
    for step in range(3):
        for wg in range(6):
            simd = 0 if wg < 5 else 1
            hw = (simd << 4) | (3 << 8) | (1 << 13)
            t0 = 1000 + step * 1000
            r.append([(4 << 56) | (step << 32) | wg, (2 << 32) | hw, t0, t0 + (500 if simd == 0 else 100)])
    a = np.array(r, dtype=np.uint64)
    path = "/tmp/zpq_trace_selftest.bin"
    a.tofile(path)
    rec = load(path)
    assert len(rec["t0"]) == 18 and len(np.unique(rec["simd_key"])) == 2
    report(rec)


if __name__ == "__main__":
    if len(sys.argv) < 2:
        selftest()
    else:
        report(load(sys.argv[1]))
