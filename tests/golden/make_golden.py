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


def random_hcomp(seed):
    """A random but terminating HCOMP program exercising the whole ZPAQL instruction set
    (SURVEY App. A.4): every operand kind, swaps, hash/hashd, R registers, div/mod (by zero too),
    shifts, comparisons, structured IF/IFNOT/ELSE, IFL/ELSEL long jumps and bounded DO loops."""
    rng = np.random.RandomState(seed)
    regs = ["a", "b", "c", "d", "*b", "*c", "*d"]

    def simple():
        k = rng.randint(0, 12)
        x = regs[rng.randint(0, 7)]
        y = regs[rng.randint(0, 7)]
        if k == 0:
            return f"{x}++" if rng.randint(2) else f"{x}--"
        if k == 1:
            return f"{x}!" if rng.randint(2) else f"{x}=0"
        if k == 2:
            return f"{regs[rng.randint(1, 7)]}<>a"
        if k == 3:
            return f"{x}={y}"
        if k == 4:
            return f"{x}= {rng.randint(0, 256)}"
        if k == 5:
            op = ["+=", "-=", "*=", "/=", "%=", "&=", "&~", "|=", "^=", "<<=", ">>="][rng.randint(0, 11)]
            return f"a{op}{y}" if rng.randint(2) else f"a{op} {rng.randint(0, 256)}"
        if k == 6:
            return "hash" if rng.randint(2) else "hashd"
        if k == 7:
            return f"r=a {rng.randint(0, 256)}"
        if k == 8:
            return f"{regs[rng.randint(0, 4)]}=r {rng.randint(0, 256)}"
        if k == 9:
            return "out"
        if k == 10:
            return f"a=c a+= {rng.randint(0, 256)} hashd"
        return f"d= {rng.randint(0, 3)} hashd"

    def cond():
        op = ["==", "<", ">"][rng.randint(0, 3)]
        return f"a{op}{regs[rng.randint(0, 7)]}" if rng.randint(2) else f"a{op} {rng.randint(0, 256)}"

    def block(depth):
        out = []
        for _ in range(rng.randint(2, 7)):
            k = rng.randint(0, 10)
            if depth < 3 and k == 0:
                out += [cond(), "if"] + block(depth + 1) + ["endif"]
            elif depth < 3 and k == 1:
                out += [cond(), "ifnot"] + block(depth + 1) + ["else"] + block(depth + 1) + ["endif"]
            elif depth < 3 and k == 2:
                out += [cond(), "ifl"] + block(depth + 1) + ["elsel"] + block(depth + 1) + ["endif"]
            elif depth < 2 and k == 3:       # bounded loop: counter in R[250+depth]
                r = 250 + depth
                out += [f"a= {rng.randint(1, 6)}", f"r=a {r}", "do"] + block(3) + \
                       [f"a=r {r}", "a--", f"r=a {r}", "a> 0", "while"]
            else:
                out.append(simple())
        return out

    body = ["c++", "*c=a", "b=c"] + block(0) + ["d=0", "hashd", "d++", "a=*c", "hashd", "d++", "a=b", "hashd", "halt"]
    return ("comp 2 4 0 0 4 (hh hm ph pm n)\n  0 cm 10 40\n  1 icm 8\n  2 isse 9 1\n  3 mix2 6 0 2 24 255\nhcomp\n  "
            + " ".join(body) + "\nend\n")


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

    # random ZPAQL programs: pins the HCOMP VM (oracle) and the HCOMP->HIP translation (GPU tests)
    out["vm_cases"] = []
    seed = 0
    while len(out["vm_cases"]) < 12:
        seed += 1
        cfg = random_hcomp(1000 + seed)
        d = mixed(1500, 500 + seed)
        try:
            a = ref.compress_config(d, cfg, None, "vm", str(seed))
        except RuntimeError:      # e.g. "IF too big": the generator does not track block sizes
            continue
        f = parse_block(a)
        out["vm_cases"].append({"name": f"random_hcomp_{seed}", "config": cfg, "gen": "mixed", "n": 1500, "seed": 500 + seed,
                                "len": len(a), "sha1": hashlib.sha1(a).hexdigest(), "header": f["header"].hex(),
                                "payload_start": f["payload_start"], "archive_b64": base64.b64encode(a).decode()})

    # more than 64 components: only the generic one-lane kernel takes these
    big = "comp 3 8 0 0 70\n" + "".join(
        (f"  {i} icm 6\n" if i % 3 == 0 else f"  {i} isse 7 {i - 1}\n") for i in range(68)) + \
        "  68 mix 8 0 68 24 255\n  69 mix2 0 68 67 24 0\n" \
        "hcomp\n  c++ *c=a b=c a=0 d=0 hash *d=a d++ b-- hash *d=a d++ b-- hash *d=a d++ a=c *d=a d++ d++ hashd halt\nend\n"
    d = mixed(3000, 321)
    a = ref.compress_config(d, big, None, "big", "3000")
    f = parse_block(a)
    out["config_cases"].append({"name": "seventy_components", "config": big, "gen": "mixed", "n": 3000, "seed": 321,
                                "len": len(a), "sha1": hashlib.sha1(a).hexdigest(), "header": f["header"].hex(),
                                "payload_start": f["payload_start"], "archive_b64": base64.b64encode(a).decode()})

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
