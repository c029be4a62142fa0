#!/usr/bin/env python3
"""Debug aid (round 6, call 22): the small-chain GPU test's exact SEQUENCE of batches in one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import zpaq_amd as z
from zpaq_amd import corpus
from oracle.oracle_py import Oracle
oracle = Oracle()
sys.path.insert(0, os.path.join(ROOT, "profiles", "r06"))
CFGS = [
    "comp 1 0 0 0 1\n  0 icm 12\nhcomp\n  *d=a halt\nend\n",
    "comp 2 0 0 0 2\n  0 icm 4\n  1 isse 4 0\nhcomp\n  b=a a=*d a<<= 4 a+=b *d=a d++ a<<= 3 a+=b *d=a halt\nend\n",
    "comp 2 3 0 0 2\n  0 icm 10\n  1 isse 10 0\nhcomp\n  c++ *c=a b=c a=0 d=0 hash b-- hash *d=a d++ b-- hash *d=a halt\nend\n",
    "comp 2 0 0 0 4\n  0 cm 9 255\n  1 icm 9\n  2 isse 10 1\n  3 isse 11 2\nhcomp\n  b=a *d=a d++ a=*d a<<= 8 a+=b *d=a d++ a<<= 2 a+=b *d=a d++ hash *d=a halt\nend\n",
]
kinds = ["text", "lcg", "zeros", "records", "pattern"]
blk = corpus.block("lcg", 1 << 18, corpus.BASE_SEED)
h3 = z.method_to_header(z.expand_method("3", blk))[0]
ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65, 2000, 777, 5000, 513, 512, 511])]
many = [b"\0" + corpus.block(kinds[i % 5], 100 + 131 * i, i).tobytes() for i in range(100)]
long_ones = [corpus.block("lcg", 40000, 100 + i).tobytes() for i in range(33)]
def run(name, hdr, inputs):
    plan = z.Plan(hdr)
    got = z.encode_batch([plan] * len(inputs), inputs)
    bad = []
    for i, (g, d) in enumerate(zip(got, inputs)):
        w = oracle.encode(hdr, d)
        if g != w:
            first = next((j for j in range(min(len(g), len(w))) if g[j] != w[j]), min(len(g), len(w)))
            bad.append((i, len(d), len(g), len(w), first))
    print(name, "persistent", z.lib().zpq_last_persistent(), "bad:", bad[:12], "of", len(bad), flush=True)
order = sys.argv[1] if len(sys.argv) > 1 else "test"
if order == "test":
    run("h3 ragged", h3, ragged); run("h3 many", h3, many); run("h3 long", h3, long_ones)
    for ci, c in enumerate(CFGS):
        run("cfg%d ragged" % ci, z.assemble(c)[0], ragged)
elif order == "long-first":
    run("h3 long", h3, long_ones)
    run("cfg0 ragged", z.assemble(CFGS[0])[0], ragged)
    run("cfg0 ragged again", z.assemble(CFGS[0])[0], ragged)
elif order == "many-first":
    run("h3 many", h3, many)
    run("cfg0 ragged", z.assemble(CFGS[0])[0], ragged)
    run("cfg0 ragged again", z.assemble(CFGS[0])[0], ragged)
