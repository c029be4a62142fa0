#!/usr/bin/env python3
"""Quick GPU probe of the pipelined encoder: bit-exactness against the oracle on a ragged mixed batch, then
kernel timings at a few batch shapes (kernel 4 = pipe, 3 = one block per wavefront)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import zpaq_amd as z
from zpaq_amd import corpus
from oracle.oracle_py import Oracle, parse_block

z.init(0)
orc = Oracle()
kinds = ["text", "lcg", "zeros", "records", "pattern"]
blocks = [corpus.block(kinds[i % 5], [0, 1, 63, 64, 65, 511, 512, 513, 5000, 20000][i % 10] + 7 * i, 100 + i) for i in range(150)]
for kernel in (4,):
    z.set_kernel(kernel)
    t0 = time.time()
    archives = z.compress_blocks(blocks, "5")
    bad = 0
    for d, a in zip(blocks, archives):
        f = parse_block(a)
        coded = orc.encode(f["header"], b"\0" + d.tobytes())
        ps = f["payload_start"]
        if a[ps:ps + len(coded) + 4] != coded + b"\0\0\0\0":
            bad += 1
    back = z.decompress(b"".join(archives))
    rt = back == b"".join(b.tobytes() for b in blocks)
    print(f"kernel {kernel}: {len(blocks)} ragged blocks, mismatches vs oracle: {bad}, round trip: {rt}, {time.time()-t0:.1f}s", flush=True)
z.set_kernel(0)
