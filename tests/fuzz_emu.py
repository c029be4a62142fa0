#!/usr/bin/env python3
"""Random MODELS through the device code on the host: a random but valid ZPAQL config (1..8 components of every type with
random sizes, inputs, rates and masks, a random HCOMP program) is assembled, and the kernels the engine would build for
that header -- the per-header wavefront coder (encode and decode, spec_kernel.h), the decoder with two blocks per
wavefront (spec_dual_kernel.h) and the pipelined encoder in its shapes (pipe_kernel.h) -- are run by the wavefront
emulator (tests/emu) on a few small blocks.  Every coded stream must be the oracle's, every decode must return the input.  No GPU; about 20 s of compilation per model.

    python tests/fuzz_emu.py [models] [seed] [--big] [--pipe-only]      (--big: 9..24 components, the 8-block workgroup shape, the
                                                           latency shapes of the pipelined encoder; --pipe-only: only the pipelined
                                                           encoder, step kernels and persistent launch)
"""
from __future__ import annotations

import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "emu"))


def random_model(rng: random.Random, big: bool = False) -> str:
    import fuzz_host
    n = rng.randrange(9, 25) if big else rng.randrange(1, 9)
    hh, hm = rng.randrange(0, 8), rng.randrange(0, 10)
    lines = ["comp %d %d 0 0 %d" % (hh, hm, n)]
    for i in range(n):
        t = rng.choice(["const", "cm", "icm", "match", "avg", "mix2", "mix", "isse", "sse"]) if i else rng.choice(["const", "cm", "icm", "match"])
        j = rng.randrange(i) if i else 0
        k = rng.randrange(i) if i else 0
        sz = rng.randrange(0, 12)
        m0 = rng.randrange(i) if i else 0
        args = {"const": [rng.randrange(256)], "cm": [sz, rng.randrange(256)], "icm": [sz], "match": [sz, rng.randrange(0, 12)],
                "avg": [j, k, rng.randrange(256)], "mix2": [sz, j, k, rng.randrange(256), rng.choice([0, 255, rng.randrange(256)])],
                "mix": [min(sz, 9), m0, rng.randrange(1, i - m0 + 1) if i else 1, rng.randrange(256), rng.choice([0, 255, rng.randrange(256)])],
                "isse": [sz, j], "sse": [sz, j, rng.randrange(0, 64), rng.randrange(64, 256)]}[t]
        lines.append("  %d %s %s" % (i, t, " ".join(map(str, args))))
    code = fuzz_host.random_code(rng, 0, False) + fuzz_host.random_code(rng, 0, False)
    if rng.random() < 0.4:              # a counted loop (backward jumps: the translators' step budget is in the path)
        code += ["b= %d" % rng.randrange(1, 30), "do"] + [w for w in fuzz_host.random_code(rng, 1, False) if not w.startswith(("b", "*b=", "a<>b"))] + \
                ["b--", "a=b", "a> 0", "while"] + fuzz_host.random_code(rng, 0, False)
    code = [w for w in code if not w.startswith(("out", "error", "lj", "jt", "jf", "jmp", "halt", "a+= $", "a= $"))]
    lines += ["hcomp"] + ["  " + " ".join(code)] + ["  halt", "end"]
    return "\n".join(lines)


def run(models: int, seed: int, verbose: bool = True, big: bool = False, pipe_only: bool = False) -> int:
    """0: every model agreed with the oracle; 1: a mismatch (printed with the config)."""
    import emu
    import zpaq_amd as z
    from oracle.oracle_py import Oracle
    from zpaq_amd import corpus
    rng = random.Random(seed)
    orc = Oracle()
    done = skipped = 0
    npersist = [0]
    t0 = time.time()
    while done < models:
        cfg = random_model(rng, big)
        try:
            header, _ = z.assemble(cfg)
            z.Plan(header)
        except z.ZpaqError:
            skipped += 1
            continue
        datas = [corpus.block("text", 700, rng.randrange(1 << 20)).tobytes(), corpus.block("lcg", 300, 3).tobytes(),
                 bytes(400), corpus.block("records", 500, rng.randrange(1 << 20)).tobytes(), b""]
        # lengths around the step size of the emulated pipeline (64 bytes), a random walk, a short period
        for _ in range(rng.randrange(0, 3)):
            n = rng.choice([1, 2, 63, 64, 65, 130, 517])
            kind = rng.randrange(3)
            if kind == 0:
                datas.append(corpus.block(rng.choice(["text", "lcg", "records"]), n, rng.randrange(1 << 20)).tobytes())
            elif kind == 1:
                v, walk = 0, bytearray()
                for _ in range(n):
                    v = (v + rng.randrange(-3, 4)) & 255
                    walk.append(v)
                datas.append(bytes(walk))
            else:
                unit = bytes(rng.randrange(256) for _ in range(7))
                datas.append((unit * (n // 7 + 1))[:n])
        inputs = [b"\0" + d for d in datas]
        want = []
        try:
            want = [orc.encode(header, i) for i in inputs]
        except Exception as ex:          # HCOMP that fails at run time (division by zero is fine in ZPAQL; jumps out of range are not)
            skipped += 1
            continue
        try:
            waves = 8 if big else 4
            if not pipe_only:
                enc = emu.run(header, inputs, waves=waves)
                for w, (coded, status, consumed), i in zip(want, enc, inputs):
                    assert status == 0 and consumed == len(i) and coded == w, ("spec encode", status, consumed, len(i))
                dec = emu.run(header, [c + b"\0\0\0\0" for c in want], decode=True, waves=waves, out_cap=max(len(x) for x in inputs))
                for i, (plain, status, consumed) in zip(inputs, dec):
                    assert status == 0 and plain == i, ("spec decode", status, len(plain), len(i))
                # the decoder with two blocks per wavefront (chains of up to 32 components whose MIX inputs fit a half)
                try:
                    emu.dual_source(header)
                    has_dual = True
                except RuntimeError:
                    has_dual = False
                if has_dual:
                    dec2 = emu.run(header, [c + b"\0\0\0\0" for c in want], decode=True, out_cap=max(len(x) for x in inputs), dual=True)
                    for i, (plain, status, consumed) in zip(inputs, dec2):
                        assert status == 0 and plain == i, ("dual decode", status, len(plain), len(i))
                # the lockstep decoder (chains whose ISSEs are fed by the ICM / ISSE before them)
                try:
                    emu.team_source(header)
                    has_team = True
                except RuntimeError:
                    has_team = False
                if has_team:
                    dec3 = emu.run(header, [c + b"\0\0\0\0" for c in want], decode=True, out_cap=max(len(x) for x in inputs), team=True)
                    for i, (plain, status, consumed) in zip(inputs, dec3):
                        assert status == 0 and plain == i, ("lockstep decode", status, len(plain), len(i))
            for mode in ((1, 2) if big else (0, 1)):
                out = emu.pipe_run(header, inputs, mode=mode, group=rng.choice([None, None, None, 8, 16]))
                for w, (coded, status, _consumed), i in zip(want, out, inputs):
                    assert status == 0 and coded == w, ("pipe mode %d" % mode, status)
            # the persistent launch (device/pipe_persist.h): the packer's plan for THIS chain -- units, dependencies, LDS regions
            # -- run as a grid of live workgroups (modes 0 and 1; the long-step shape has none)
            for mode in (0, 1):
                if "PS_WPG" not in emu.pipe_source(header, 64, mode=mode):
                    continue
                out = emu.pipe_run(header, inputs, mode=mode, chunk=64, group=rng.choice([None, 8, 16]), persist=True)
                for w, (coded, status, _consumed), i in zip(want, out, inputs):
                    assert status == 0 and coded == w, ("persistent launch, mode %d" % mode, status)
                npersist[0] += 1
        except AssertionError as ex:
            print("MISMATCH", ex.args, "\n" + cfg, flush=True)
            return 1
        except RuntimeError as ex:
            print("BUILD/RUN FAILURE", str(ex)[-1500:], "\n" + cfg, flush=True)
            return 1
        done += 1
        if verbose:
            print("model %d ok (%d comps, %.0f s)" % (done, header[6], time.time() - t0), flush=True)
    if verbose:
        print("models", done, "skipped", skipped, "persistent launches run", npersist[0])
    return 0


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    return run(int(pos[0]) if pos else 10, int(pos[1]) if len(pos) > 1 else 1, big="--big" in sys.argv, pipe_only="--pipe-only" in sys.argv)


if __name__ == "__main__":
    sys.exit(main())
