#!/usr/bin/env python3
"""Compiles only the code objects the bench's headline needs (the -m5 chain of 1 MiB blocks, the pipelined encoder in its
shapes) into zpaq_amd/spec_cache -- for GPU calls between two full prebuilds (python -m zpaq_amd.prebuild takes much longer)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import zpaq_amd as z
from zpaq_amd import prebuild, corpus
from concurrent.futures import ThreadPoolExecutor
L = z.lib()
L.zpq_spec_cache_dir.restype = C.c_char_p; L.zpq_spec_include_dir.restype = C.c_char_p
cache = L.zpq_spec_cache_dir().decode(); inc = L.zpq_spec_include_dir().decode()
jobs = []; seen = set(); hs = set()
sizes = [int(x) for x in sys.argv[1:]] or [1 << 20]
for n in sizes:
    for kind in ("text", "lcg", "zeros", "pattern"):
        hs.add(z.method_to_header(z.expand_method("5", corpus.block(kind, n, 5)))[0])
for h in hs:
    for mode in (0, 1):
        src, key = prebuild.pipe_source_and_key(h, mode)
        if src and key not in seen:
            seen.add(key); jobs.append((src, key, cache, inc))
with ThreadPoolExecutor(max_workers=8) as ex:
    print(list(ex.map(prebuild.compile_one, jobs)))
