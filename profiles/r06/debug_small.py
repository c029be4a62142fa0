#!/usr/bin/env python3
"""Debug aid (round 6, calls 20-21): the small-chain GPU test's batches, block by block against the oracle, under the
code-generation knobs of the environment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import zpaq_amd as z
from zpaq_amd import corpus
from oracle.oracle_py import Oracle
oracle = Oracle()
CFGS = [
    "comp 1 0 0 0 1\n  0 icm 12\nhcomp\n  *d=a halt\nend\n",
    "comp 2 0 0 0 2\n  0 icm 4\n  1 isse 4 0\nhcomp\n  b=a a=*d a<<= 4 a+=b *d=a d++ a<<= 3 a+=b *d=a halt\nend\n",
    "comp 2 3 0 0 2\n  0 icm 10\n  1 isse 10 0\nhcomp\n  c++ *c=a b=c a=0 d=0 hash b-- hash *d=a d++ b-- hash *d=a halt\nend\n",
    "comp 2 0 0 0 4\n  0 cm 9 255\n  1 icm 9\n  2 isse 10 1\n  3 isse 11 2\nhcomp\n  b=a *d=a d++ a=*d a<<= 8 a+=b *d=a d++ a<<= 2 a+=b *d=a d++ hash *d=a halt\nend\n",
]
kinds = ["text", "lcg", "zeros", "records", "pattern"]
sizes = [300, 150, 200, 97, 0, 1, 63, 64, 65, 2000, 777, 5000, 513, 512, 511]
ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate(sizes)]
def run(name, plan, hdr, inputs):
    got = z.encode_batch([plan] * len(inputs), inputs)
    bad = []
    for i, (g, d) in enumerate(zip(got, inputs)):
        w = oracle.encode(hdr, d)
        if g != w:
            first = next((j for j in range(min(len(g), len(w))) if g[j] != w[j]), min(len(g), len(w)))
            bad.append((i, len(d), len(g), len(w), first))
    print(name, "persistent", z.lib().zpq_last_persistent(), "bad (block, in_len, got_len, want_len, first diff):", bad, flush=True)
which = [int(x) for x in sys.argv[1:]] or range(len(CFGS))
for ci in which:
    hdr = z.assemble(CFGS[ci])[0]
    plan = z.Plan(hdr)
    run("cfg%d ragged15" % ci, plan, hdr, ragged)
    run("cfg%d ragged15 again" % ci, plan, hdr, ragged)
    run("cfg%d block0 alone" % ci, plan, hdr, ragged[:1])
    run("cfg%d equal 32 x 300 text" % ci, plan, hdr, [b"\0" + corpus.block("text", 300, 40 + i).tobytes() for i in range(32)])
    run("cfg%d 8 x 3000 lcg" % ci, plan, hdr, [b"\0" + corpus.block("lcg", 3000, 40 + i).tobytes() for i in range(8)])
