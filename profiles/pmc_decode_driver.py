#!/usr/bin/env python3
"""Torch-free driver for PMC passes over the DECODER (rocprofv3 --pmc crashes inside processes that import torch on this
image).  The decoder is one dispatch, so -- unlike the encoder's 12 384 -- a counter pass over it finishes.

    python profiles/pmc_decode_driver.py make 2048 1048576 131072 /tmp/zpq_dec.npz     (no profiler: codes the inputs)
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- python profiles/pmc_decode_driver.py run /tmp/zpq_dec.npz

make: the first PREFIX bytes of N Zipf-text blocks, coded with the chain compressBlock gives BS-byte blocks (method 5): the
decoder's tables, chain and code object of the bench's decode leg; a block's traffic per byte does not depend on how far into
the block it is (every table access is a random line at any fill level).  run: init_arena_kernel + zpq_spec_decode3 (kernel
choice 6: the lockstep decoder), every decoded byte compared with the input."""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def gen(args):
    from zpaq_amd import corpus
    return corpus.block(args[2] if len(args) > 2 else "text", args[0], corpus.BASE_SEED + args[1])


def main():
    import zpaq_amd as z
    from zpaq_amd import corpus
    if sys.argv[1] == "make":
        nb, bs, prefix, path = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
        kind = sys.argv[6] if len(sys.argv) > 6 else "text"      # "records": the n = 29 chain of the mixed corpus (round 6)
        first = corpus.block(kind, bs, corpus.BASE_SEED)
        hdr = z.method_to_header(z.expand_method("5", first))[0]
        with mp.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:
            blocks = pool.map(gen, [(prefix, b, kind) for b in range(nb)], chunksize=8)
        z.init(0)
        plan = z.Plan(hdr)
        print("chain: n =", plan.ncomp, "algorithmic bytes per byte", plan.algo_bytes_per_byte)
        coded = z.encode_batch([plan] * nb, [b"\0" + b.tobytes() for b in blocks])
        lens = np.array([len(c) for c in coded], np.int64)
        np.savez(path, header=np.frombuffer(bytes(hdr), np.uint8), lens=lens, coded=np.frombuffer(b"".join(coded), np.uint8),
                 plain=np.concatenate(blocks), prefix=np.array([prefix]))
        print("coded", nb, "x", prefix, "of", bs, "->", int(lens.sum()), "bytes")
        return
    d = dict(np.load(sys.argv[2]))            # (an NpzFile reads an array from the file at EVERY access: load them once)
    hdr, lens, prefix = d["header"].tobytes(), d["lens"], int(d["prefix"][0])
    offs = np.concatenate([[0], np.cumsum(lens)])
    all_coded, plain = d["coded"], d["plain"]
    coded = [all_coded[offs[i]:offs[i + 1]] for i in range(len(lens))]
    nb = len(coded)
    z.init(0)
    z.set_kernel(6)
    plan = z.Plan(hdr)
    t0 = time.time()
    out = z.decode_batch([plan] * nb, coded, [prefix + 1] * nb)
    ok = all(o[0][1:] == plain[i * prefix:(i + 1) * prefix].tobytes() for i, o in enumerate(out))
    print("decoded", nb, "x", prefix, "in %.2f s" % (time.time() - t0), "kernel ms", z.last_timing(), "output bytes", nb * (prefix + 1), "identical", ok)


if __name__ == "__main__":
    main()
