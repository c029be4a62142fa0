#!/usr/bin/env python3
"""Generates tests/golden/golden.json from the REFERENCE itself (oracle/_ref,
built from /root/reference by oracle/Makefile).  Run in the container that has
/root/reference:   python tests/golden/make_golden.py

Every entry pins what unmodified libzpaq 7.15 produces for a deterministic
input (zpaq_amd.corpus generators), so the -m "not gpu" tests can check the C
oracle and the -m gpu tests can check the HIP path on a box without the
reference sources.
"""
import base64
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle.oracle_py import Ref, parse_block  # noqa: E402
from zpaq_amd import corpus  # noqa: E402

# A hand-written config using all nine component types (SURVEY App. A validation set).
ALL_TYPES_CFG = """
comp 4 12 0 0 10 (hh hm ph pm n)
  0 const 160
  1 cm 12 20
  2 icm 10
  3 match 12 14
  4 avg 1 2 100
  5 isse 11 2
  6 mix2 8 4 5 24 255
  7 mix 10 0 7 16 255
  8 sse 10 7 16 255
  9 mix2 0 8 6 20 0
hcomp
  c++ *c=a b=c a=0 (save in rotating buffer M)
  d= 1 hash *d=a (orders 1,2 for cm, icm)
  b-- d++ hash *d=a
  b-- d++ hash *d=a (order 3 match)
  d= 5 a=*d d= 5 b=c hash b-- hash b-- hash b-- hash *d=a (order 4 isse)
  d= 6 a=*c a>>= 4 *d=a (mix2 context: high nibble)
  d= 7 a=*c a<<= 2 *d=a
  d= 8 a=*c a<<= 1 *d=a
  halt
end
"""


def mixed(n, seed):
    parts = [corpus.block(k, n // 4 + 1, seed + i) for i, k in enumerate(["text", "lcg", "zeros", "records"])]
    return np.concatenate(parts)[:n]


def main():
    ref = Ref()
    out = {"_generator": "tests/golden/make_golden.py", "_reference": "zpaq 7.15 libzpaq (oracle/_ref, JIT build)",
           "known_answers": [], "method_cases": [], "config_cases": [], "level_cases": []}

    # BASELINE.md §2 known answers (generator independent: zeros / the LCG stream)
    for kind, n, method in [("zeros", 65536, "1"), ("zeros", 65536, "5"), ("zeros", 1 << 20, "5"),
                            ("lcg", 262144, "3"), ("lcg", 1 << 20, "5")]:
        a = ref.compress_block(corpus.block(kind, n, corpus.BASE_SEED), method)
        out["known_answers"].append({"kind": kind, "n": n, "seed": corpus.BASE_SEED, "method": method,
                                     "len": len(a), "sha1": hashlib.sha1(a).hexdigest()})

    cases = []
    for method in ["5", "4"]:
        for kind in ["text", "lcg", "zeros", "records", "pattern"]:
            for n in [0, 1, 2, 777, 20000, 65536]:
                cases.append((kind, n, method, None, None))
    cases += [("text", 262144, "5", None, None), ("records", 262144, "5", None, None),
              ("text", 100000, "5,128,1", "name.txt", "a comment"), ("lcg", 50000, "5,250,0", None, None),
              ("text", 50000, "4,128,1", "x", None), ("text", 30000, "0", None, None),
              ("mixed", 150000, "5", None, None), ("text", 3000, "9", "f", "c")]
    for kind, n, method, fn, cm in cases:
        d = mixed(n, corpus.BASE_SEED) if kind == "mixed" else corpus.block(kind, n, corpus.BASE_SEED)
        a = ref.compress_block(d, method, fn, cm)
        f = parse_block(a)
        e = {"kind": kind, "n": n, "seed": corpus.BASE_SEED, "method": method, "filename": fn, "comment": cm,
             "len": len(a), "sha1": hashlib.sha1(a).hexdigest(), "header": f["header"].hex(),
             "payload_start": f["payload_start"], "memory": ref.block_memory(a)}
        if len(a) <= 2048:
            e["archive_b64"] = base64.b64encode(a).decode()
        out["method_cases"].append(e)

    d = mixed(39000, 99)
    a = ref.compress_config(d, ALL_TYPES_CFG, None, "all9", "39000")
    f = parse_block(a)
    out["config_cases"].append({"name": "all_nine_types", "config": ALL_TYPES_CFG, "gen": "mixed", "n": 39000,
                                "seed": 99, "len": len(a), "sha1": hashlib.sha1(a).hexdigest(),
                                "header": f["header"].hex(), "payload_start": f["payload_start"],
                                "archive_b64": base64.b64encode(a).decode()})

    for level in (1, 2, 3):   # legacy built-in min/mid/max.cfg (libzpaq.cpp:2793-2831)
        d = mixed(20000, 7 + level)
        a = ref.compress_level(d, level, "lvl", "20000")
        f = parse_block(a)
        out["level_cases"].append({"level": level, "gen": "mixed", "n": 20000, "seed": 7 + level, "len": len(a),
                                   "sha1": hashlib.sha1(a).hexdigest(), "header": f["header"].hex(),
                                   "payload_start": f["payload_start"],
                                   "archive_b64": base64.b64encode(a).decode()})

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
