#!/usr/bin/env python3
"""Compile the pipelined encoder of the headline chain (-m5, 1 MiB text blocks: x1) in the given modes for each
ZPAQ_AMD_SPEC_DEFS variant of an A/B ahead of the GPU call:  prebuild_pipe_variants.py <modes, e.g. 012> <defs|none>..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import zpaq_amd as z
from zpaq_amd import corpus, prebuild

def main():
    L = z.lib()
    L.zpq_spec_cache_dir.restype = C.c_char_p
    L.zpq_spec_include_dir.restype = C.c_char_p
    cache, inc = L.zpq_spec_cache_dir().decode(), L.zpq_spec_include_dir().decode()
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    h = z.method_to_header(z.expand_method("5", blk))[0]
    modes = [int(c) for c in sys.argv[1]]
    for defs in sys.argv[2:]:
        if defs == "none":
            os.environ.pop("ZPAQ_AMD_SPEC_DEFS", None)
        else:
            os.environ["ZPAQ_AMD_SPEC_DEFS"] = defs
        for mode in modes:
            src, key = prebuild.pipe_source_and_key(h, mode)
            print(defs, mode, key, prebuild.compile_one((src, key, cache, inc))[1], flush=True)

if __name__ == "__main__":
    main()
