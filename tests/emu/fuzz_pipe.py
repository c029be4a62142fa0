"""TEST INFRASTRUCTURE -- random COMP chains and HCOMP programs through the pipelined encoder under the wavefront emulator,
with every experimental unit switched on (bit-lane MIX / CM / MIX2 / SSE, nibble-lane ROW units, random fetch depths), compared
with the oracle byte for byte.  Not collected by pytest (minutes of g++): run by hand after touching pipe_kernel.h --
    python tests/emu/fuzz_pipe.py <seed> <cases> [default]       (600 chains passed at the end of round 2; `default` = knobs off,
                                                                  `wavefront` = spec_kernel.h, encoder and decoder)"""
import sys, random, numpy as np, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")]
import zpaq_amd as z
from zpaq_amd import corpus
import emu
from oracle.oracle_py import Oracle
oracle = Oracle()
seed0 = int(sys.argv[1]); ncase = int(sys.argv[2])
HC = ["c++ *c=a b=c", "d= 0 *d=a", "d++ a=*b a>>= 1 *d=a", "d++ a=*b a<<= 1 *d=a", "d++ b-- a=*b a+=*c *d=a", "d++ hash *d=a",
      "d++ a=*c a>>= 6 *d=a", "d++ a=*c a>>= 2 *d=a", "d++ a=*c a>>= 3 *d=a", "d++ a=*c a<<= 1 *d=a", "d++ a=*c a<<= 8 *d=a",
      "d++ a= 0 *d=a", "d++ a=*c a&= 3 *d=a", "d++ hashd", "d++ a=*b hashd"]
def make_cfg(r):
    n = r.randint(3, 12)
    comps = []
    for i in range(n):
        choices = ["cm", "icm"]
        if i >= 1: choices += ["isse", "sse", "avg"] if i >= 2 else ["isse", "sse"]
        if i >= 2: choices += ["mix2", "mix", "mix", "mix2"]
        choices += ["match"] if r.random() < 0.3 else []
        t = r.choice(choices)
        if i == n - 1 and i >= 2: t = r.choice(["mix", "mix2", "sse"])
        if t == "cm": comps.append(f"cm {r.choice([9, 10, 12, 16])} {r.choice([4, 20, 255])}")
        elif t == "icm": comps.append(f"icm {r.choice([1, 2, 4, 8])}")
        elif t == "isse": comps.append(f"isse {r.choice([1, 2, 5, 9])} {r.randrange(i)}")
        elif t == "sse": comps.append(f"sse {r.choice([8, 9, 12])} {r.randrange(i)} {r.choice([4, 32])} {r.choice([32, 255])}")
        elif t == "avg": a = r.randrange(i); b = r.randrange(i); comps.append(f"avg {a} {b} {r.choice([64, 128, 200])}")
        elif t == "mix2": a = r.randrange(i); b = r.randrange(i); comps.append(f"mix2 {r.choice([0, 8, 9, 12])} {a} {b} {r.choice([8, 24])} {r.choice([0, 255, 255, 15])}")
        elif t == "mix":
            j = r.randrange(i); m = r.randint(1, min(i - j, 20))
            comps.append(f"mix {r.choice([8, 8, 10, 16])} {j} {m} {r.choice([8, 24])} 255")
        elif t == "match": comps.append(f"match {r.choice([8, 10])} {r.choice([10, 12])}")
    hh = 0
    while (1 << hh) < n: hh += 1
    lines = [f"comp {max(hh,1)} 8 0 0 {n}"] + [f"  {i} {c}" for i, c in enumerate(comps)] + ["hcomp"]
    prog = [HC[0], HC[1]] + [r.choice(HC[2:]) for _ in range(n - 1)]
    lines += ["  " + p for p in prog] + ["  halt", "end"]
    return "\n".join(lines)
def make_data(r, k):
    kind = r.choice(["text", "lcg", "zeros", "records", "pattern", "walk", "rep"])
    n = r.choice([1, 2, 63, 64, 65, 130, 300, 517])
    if kind == "walk":
        g = np.random.default_rng(k); return (np.cumsum(g.integers(-3, 4, n)) & 255).astype(np.uint8).tobytes()
    if kind == "rep":
        g = np.random.default_rng(k); return (bytes(g.integers(0, 256, 7, dtype=np.uint8)) * (n // 7 + 1))[:n]
    return corpus.block(kind, n, k).tobytes()
bad = 0; done = 0; t0 = time.time()
for case in range(ncase):
    r = random.Random(seed0 * 1000 + case)
    cfg = make_cfg(r)
    try:
        header, _ = z.assemble(cfg)
    except Exception as e:
        continue
    inputs = [b"\0" + make_data(r, case * 10 + i) for i in range(r.choice([1, 3, 5]))] + ([b""] if r.random() < 0.3 else [])
    kw = dict(chunk=64, mode=r.choice([0, 1]))
    if len(sys.argv) > 3 and sys.argv[3] == "wavefront":
        # the per-header wavefront kernel (spec_kernel.h) in both directions instead: encode == oracle, decode(oracle) == input
        waves = r.choice([4, 8])
        try:
            enc = emu.run(header, inputs, decode=False, waves=waves)
            coded = [oracle.encode(header, x) for x in inputs]
            dec = emu.run(header, [cd + b"\0\0\0\0" for cd in coded], decode=True, waves=waves, out_cap=max(len(x) for x in inputs) + 8)
        except Exception as e:
            print("CASE", seed0, case, "emulator error:", str(e)[-300:]); bad += 1; continue
        ok = all(e[0] == cd and e[1] == 0 for e, cd in zip(enc, coded)) and all(d[0] == x and d[1] == 0 for d, x in zip(dec, inputs))
        if not ok: print("CASE", seed0, case, "MISMATCH (wavefront kernel)"); print(cfg)
        bad += not ok; done += 1
        continue
    if r.random() < 0.2: kw["group"] = r.choice([8, 16])
    try:
        res = emu.pipe_run(header, inputs, **kw)
    except Exception as e:
        print("CASE", seed0, case, "emulator error:", str(e)[-300:]); bad += 1; continue
    ok = True
    for i, (inp, (coded, st, used)) in enumerate(zip(inputs, res)):
        if not (st == 0 and used == len(inp) and coded == oracle.encode(header, inp)):
            ok = False; print("CASE", seed0, case, "MISMATCH block", i, "len", len(inp), "status", st, kw); print(cfg)
            break
    bad += not ok; done += 1
print(f"seed {seed0}: {done} cases, {bad} bad, {time.time() - t0:.0f} s")
