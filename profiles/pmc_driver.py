#!/usr/bin/env python3
"""Torch-free driver for the PMC passes (rocprofv3 --pmc crashes inside processes that import torch on this image):
compresses N synthetic text blocks of S bytes with method 5 through zpq_compress_blocks, nothing else on the GPU.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -o p -- python profiles/pmc_driver.py 1024 1048576"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def gen(args):
    from zpaq_amd import corpus
    return corpus.block("text", args[0], corpus.BASE_SEED + args[1])


def main():
    nb, bs = int(sys.argv[1]), int(sys.argv[2])
    prefix = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if prefix:
        # SAMPLED: the chain, tables and code object of bs-byte blocks (the header compressBlock makes for them), but only the
        # first `prefix` bytes of every block are coded -- rocprofv3 serialises every dispatch it counts, and the whole
        # sequence (12 384 dispatches for 1024 x 1 MiB) does not finish in any budget.  Traffic per input byte of the
        # sample is what the whole sequence does per byte: every table access is a random line at any fill level.
        from zpaq_amd import corpus
        import zpaq_amd as z
        first = corpus.block("text", bs, corpus.BASE_SEED)
        hdr = z.method_to_header(z.expand_method("5", first))[0]
        with mp.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:
            blocks = pool.map(gen, [(prefix, b) for b in range(nb)], chunksize=8)      # (a Zipf text block's prefix = the shorter block of the same seed)
        z.init(0)
        plan = z.Plan(hdr)
        t0 = time.time()
        coded = z.encode_batch([plan] * nb, [b"\0" + b.tobytes() for b in blocks])
        print("compressed", nb, "x", prefix, "of", bs, "in %.2f s" % (time.time() - t0), "->", sum(len(a) for a in coded), "bytes",
              "kernel ms", z.last_timing(), "input bytes", nb * (prefix + 1))
        return
    with mp.get_context("fork").Pool(min(32, os.cpu_count() or 1)) as pool:
        blocks = pool.map(gen, [(bs, b) for b in range(nb)], chunksize=4)
    import zpaq_amd as z
    z.init(0)
    t0 = time.time()
    arch = z.compress_blocks(blocks, "5")
    print("compressed", nb, "x", bs, "in %.2f s" % (time.time() - t0), "->", sum(len(a) for a in arch), "bytes",
          "kernel ms", z.last_timing())


if __name__ == "__main__":
    main()
