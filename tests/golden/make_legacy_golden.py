#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Freezes what the REFERENCE (oracle/_ref: the unmodified libzpaq compiled where it lies) makes of
BASELINE configs[1] / configs[2]'s corpora with its LEGACY built-in models -- Compressor::startBlock(2) = mid.cfg over
256 x 256 KiB LCG blocks, Compressor::startBlock(3) = max.cfg over 1024 x 1 MiB Zipf text blocks (libzpaq.cpp:2793-2839;
SURVEY 8(d) C2 / C3 "also run startBlock(2) / startBlock(3) on the same data"):

    python tests/golden/make_legacy_golden.py [threads]      ->  tests/golden/legacy_sha1.json

Per block: the coded payload's length and the SHA-1 of payload + 4-zero terminator (what the device-resident encoder of
`bench.py --legacy-level L` writes) and the SHA-1 of the whole archive.  Nothing of this library or of the oracle's
restatement is involved.  Needs /root/reference (to have built oracle/_ref); level 3 costs ~2.5 s of one core per block."""
import hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zpaq_amd import corpus
from oracle.oracle_py import Ref, parse_block

CASES = {"2": ("lcg", 256, 1 << 18), "3": ("text", 1024, 1 << 20)}


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 1)
    ref = Ref()
    out = {"reference": "oracle/_ref " + ref.build_flags(), "levels": {}}
    for level, (kind, nb, bs) in CASES.items():
        ent = {"corpus": f"zpaq_amd.corpus.block('{kind}', {bs}, BASE_SEED + b), b = 0 .. {nb - 1}", "level": int(level),
               "block_bytes": bs, "blocks": []}
        t0 = time.time()
        for b0 in range(0, nb, 64):
            n = min(64, nb - b0)
            blocks = np.stack([corpus.block(kind, bs, corpus.BASE_SEED + b0 + i) for i in range(n)])
            _, lens, arch = ref.compress_blocks_mt(blocks, "L" + level, threads, keep=True)
            for a in arch:
                f = parse_block(a)
                ps = f["payload_start"]
                assert a[-22] == 253 and a[-1] == 255 and a[-26:-22] == b"\0\0\0\0"       # terminator, SHA-1 trailer, end of block
                ent["blocks"].append({"len": len(a), "sha1": hashlib.sha1(a).hexdigest(), "coded_len": len(a) - 26 - ps,
                                      "payload_sha1": hashlib.sha1(a[ps:len(a) - 22]).hexdigest()})
            print(f"level {level}: {b0 + n} / {nb} blocks, {time.time() - t0:.0f} s", flush=True)
        out["levels"][level] = ent
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "legacy_sha1.json"), "w"), indent=0)


main()
