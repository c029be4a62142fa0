#!/usr/bin/env python3
"""Builds the code object of ONE chain's pipelined encoder under the code-generation knobs of the environment
(ZPAQ_AMD_MIX_PACKED, ZPAQ_AMD_MIX_LDS_ROWS, ...) into zpaq_amd/spec_cache/ beside the product's, so that an A/B on the GPU box
does not spend its minutes in hipRTC.  The knobs are part of the generated text, hence of the cache key.

    ZPAQ_AMD_MIX_PACKED=0 python profiles/r06/prebuild_variant.py [method=5] [mode=0] [kind=text] [block_bytes=1048576]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import zpaq_amd as z
from zpaq_amd import corpus, prebuild

method = sys.argv[1] if len(sys.argv) > 1 else "5"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kind = sys.argv[3] if len(sys.argv) > 3 else "text"
bs = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 20
if method.startswith("L"):
    h = z.builtin_model_header(int(method[1:]))
else:
    h = z.method_to_header(z.expand_method(method, corpus.block(kind, bs, corpus.BASE_SEED)))[0]
if mode == 3:       # the lockstep decoder of the chain (ZPAQ_AMD_TEAM_TAIL=0: without the tail wavefront)
    src, key = prebuild.team_source_and_key(h)
else:
    src, key = prebuild.pipe_source_and_key(h, mode)
assert src is not None, key
L = z.lib()
L.zpq_spec_cache_dir.restype = C.c_char_p
L.zpq_spec_include_dir.restype = C.c_char_p
cache, inc = L.zpq_spec_cache_dir().decode(), L.zpq_spec_include_dir().decode()
os.makedirs(cache, exist_ok=True)
print(prebuild.compile_one((src, key, cache, inc)), {k: v for k, v in os.environ.items() if k.startswith("ZPAQ_AMD_")})
