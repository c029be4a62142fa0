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
