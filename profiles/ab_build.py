#!/usr/bin/env python3
"""Builds experiment variants of the headline kernel (method "5", 1 MiB and 64 KiB Zipf text) into
zpaq_amd/spec_cache so that an A/B run on the GPU box needs no hipRTC:   python profiles/ab_build.py

Variants = workgroup shapes 4/8/12/16 x {default, -DZPQ_TOUCH2=1}.  At run time select one with
ZPAQ_AMD_SPEC_WAVES=<w> and ZPAQ_AMD_SPEC_DEFS=<def> (both enter the cache key); see profiles/ab_run.sh.
Run `python -m zpaq_amd.prebuild` with ZPAQ_AMD_KEEP_CACHE=1 afterwards if you want to keep them."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import zpaq_amd as z
    from zpaq_amd import corpus, prebuild
    L = z.lib()
    L.zpq_spec_cache_dir.restype = C.c_char_p
    L.zpq_spec_include_dir.restype = C.c_char_p
    cache, inc = L.zpq_spec_cache_dir().decode(), L.zpq_spec_include_dir().decode()
    os.makedirs(cache, exist_ok=True)
    headers = set()
    for n in (1 << 20, 65536):
        blk = corpus.block("text", n, corpus.BASE_SEED)
        headers.add(z.method_to_header(z.expand_method("5", blk))[0])
    for defs in ("", "-DZPQ_TOUCH2=1"):
        if defs:
            os.environ["ZPAQ_AMD_SPEC_DEFS"] = defs
        else:
            os.environ.pop("ZPAQ_AMD_SPEC_DEFS", None)
        for w in ("4", "8", "12", "16"):
            os.environ["ZPAQ_AMD_SPEC_WAVES"] = w
            for h in headers:
                src, key = prebuild.source_and_key(h)
                print(defs or "(default)", "waves", w, *prebuild.compile_one((src, key, cache, inc)))
    # the experimental two-blocks-per-wavefront kernel, 4 and 8 wavefronts per workgroup (8 / 16 blocks)
    os.environ.pop("ZPAQ_AMD_SPEC_DEFS", None)
    os.environ["ZPAQ_AMD_SPEC_DUAL"] = "1"
    for w in ("4", "8"):
        os.environ["ZPAQ_AMD_SPEC_WAVES"] = w
        for h in headers:
            src, key = prebuild.source_and_key(h)
            print("dual", "waves", w, *prebuild.compile_one((src, key, cache, inc)))


if __name__ == "__main__":
    main()
