#!/usr/bin/env python3
"""The persistent launch of the pipelined encoder against the step kernels on the same inputs (archives must be identical),
and against the oracle on the small ones.  python profiles/r05/persist_check.py [nblocks]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import zpaq_amd as z
from zpaq_amd import corpus

def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    z.init(0)
    L = z.lib()
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    rng = np.random.default_rng(7)
    sizes = [int(x) for x in rng.integers(0, 200_000, nb)]
    sizes[:6] = [0, 1, 511, 512, 513, 70_000]
    blocks = [corpus.block(kinds[i % 5], n, 100 + i) for i, n in enumerate(sizes)]
    out = {}
    for tag, env in (("persist", None), ("steps", "0")):
        if env is None: os.environ.pop("ZPAQ_AMD_PIPE_PERSIST", None)
        else: os.environ["ZPAQ_AMD_PIPE_PERSIST"] = env
        t = time.time()
        out[tag] = z.compress_blocks(blocks, "5")
        print(tag, "%.2f s" % (time.time() - t), "persistent:", L.zpq_last_persistent(), "timing", z.last_timing(), flush=True)
    same = all(a == b for a, b in zip(out["persist"], out["steps"]))
    print("archives identical:", same)
    back = z.decompress(b"".join(out["persist"][:12]))
    print("round trip of 12:", back == b"".join(b.tobytes() for b in blocks[:12]))
    from oracle.oracle_py import Oracle, parse_block
    orc = Oracle()
    ok = True
    for i in range(min(nb, 8)):
        if sizes[i] > 80_000: continue
        a = out["persist"][i]
        f = parse_block(a)
        coded = orc.encode(f["header"], b"\0" + blocks[i].tobytes())
        ps = f["payload_start"]
        ok = ok and a[ps:ps + len(coded)] == coded
    print("oracle:", ok)
    sys.exit(0 if same and ok else 1)

main()
