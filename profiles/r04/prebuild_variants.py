#!/usr/bin/env python3
"""Code objects of the lockstep decoder for the bench's -m5 chain (1 MiB blocks, text) under a list of -D sets, built here
so that an A/B on the GPU box does not spend its minutes in hipRTC:  python profiles/r04/prebuild_variants.py "-DA" "-DB -DC" ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
from zpaq_amd import prebuild
import zpaq_amd as z

xm = z.expand_method("5,128,1", bytes(1))
xm = "x1" + xm[xm.index(","):]          # bench.py's blocks: 1 MiB -> arg0 = 1 ... see zpaq_amd.expand_method
L = z.lib()
L.zpq_spec_cache_dir.restype = C.c_char_p
L.zpq_spec_include_dir.restype = C.c_char_p
cache, inc = L.zpq_spec_cache_dir().decode(), L.zpq_spec_include_dir().decode()
for defs in sys.argv[1:]:
    os.environ["ZPAQ_AMD_SPEC_DEFS"] = defs
    for hint in ("", ",128,1"):
        xm = z.expand_method("5" + hint, bytes(1))
        xm = "x1" + xm[xm.index(","):]
        h, _, _ = z.method_to_header(xm)
        src, key = prebuild.team_source_and_key(h)
        print(defs, "|", "5" + hint, key, prebuild.compile_one((src, key, cache, inc))[1])
