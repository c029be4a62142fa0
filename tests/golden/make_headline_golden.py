#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Freezes what the REFERENCE (oracle/_ref: the unmodified libzpaq compiled where it lies) makes of the
bench's headline corpus -- BASELINE configs[2]: 1024 blocks of 1 MiB Zipf text, compressBlock method "5" -- so that every block
of the timed run can be checked against something this library had no part in, without the CPU time at bench time:

    python tests/golden/make_headline_golden.py [nblocks] [threads]      ->  tests/golden/headline_sha1.json

Per block: the archive's length, the SHA-1 of the whole archive (what zpq_compress_blocks must return byte for byte) and the
SHA-1 of the coded payload + its 4-zero terminator (what the device-resident encoder of the bench's timed region writes).
Needs /root/reference (to have built oracle/_ref); ~1.7 s of one core per block."""
import hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zpaq_amd import corpus
from oracle.oracle_py import Ref, parse_block


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    bs = 1 << 20
    ref = Ref()
    out = {"corpus": "zpaq_amd.corpus.block('text', 1048576, BASE_SEED + b), b = 0 .. nblocks - 1", "method": "5", "block_bytes": bs,
           "reference": "oracle/_ref " + ref.build_flags(), "blocks": []}
    t0 = time.time()
    for b0 in range(0, nb, 64):
        n = min(64, nb - b0)
        blocks = np.stack([corpus.block("text", bs, corpus.BASE_SEED + b0 + i) for i in range(n)])
        _, lens, arch = ref.compress_blocks_mt(blocks, "5", threads, keep=True)
        for a in arch:
            f = parse_block(a)
            ps = f["payload_start"]
            assert a[-22] == 253 and a[-1] == 255 and a[-26:-22] == b"\0\0\0\0"       # terminator, SHA-1 trailer, end of block
            out["blocks"].append({"len": len(a), "sha1": hashlib.sha1(a).hexdigest(), "coded_len": len(a) - 26 - ps,
                                  "payload_sha1": hashlib.sha1(a[ps:len(a) - 22]).hexdigest()})
        print(f"{b0 + n} / {nb} blocks, {time.time() - t0:.0f} s", flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "headline_sha1.json"), "w"), indent=0)


main()
