#!/usr/bin/env python3
"""Damaged-input fuzzing of the host-only paths of the library (no GPU): the archive container and the stored / LZ77 /
BWT blocks that are decoded on the host (PCOMP post-processors included), the block-header parser, the ZPAQL assembler
and the method-string expander.  Nothing here may crash, hang or read out of bounds: every outcome is a ZpaqError or a
result.  Meant to run against the sanitizer build:

    make -C zpaq_amd/csrc ASAN=1
    LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so):$(gcc -print-file-name=libstdc++.so) \
      ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
      ZPAQ_AMD_LIB=libzpaq_amd_asan.so python tests/fuzz_host.py [iterations] [seed]

tests/test_host.py runs a short round of it against the normal build on every CPU test run.
"""
from __future__ import annotations

import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mutate(rng: random.Random, a: bytes) -> bytes:
    b = bytearray(a)
    k = rng.randrange(6)
    if k == 0 and b:                         # flip a few bytes
        for _ in range(rng.randrange(1, 6)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    elif k == 1 and b:                       # truncate
        del b[rng.randrange(len(b)):]
    elif k == 2 and b:                       # delete a run
        i = rng.randrange(len(b))
        del b[i:i + rng.randrange(1, 40)]
    elif k == 3:                             # insert noise
        i = rng.randrange(len(b) + 1)
        b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 40)))
    elif k == 4 and b:                       # bit flips near the front (headers, PCOMP program)
        for _ in range(rng.randrange(1, 4)):
            i = rng.randrange(min(len(b), 300))
            b[i] ^= 1 << rng.randrange(8)
    elif b:                                  # repeat a slice (duplicated segments / blocks)
        i = rng.randrange(len(b))
        j = min(len(b), i + rng.randrange(1, 400))
        b[j:j] = b[i:j]
    return bytes(b)


_REG = ["a", "b", "c", "d", "*b", "*c", "*d"]
_BIN = ["+=", "-=", "*=", "/=", "%=", "&=", "&~", "|=", "^=", "<<=", ">>=", "==", "<", ">"]


def random_code(rng: random.Random, depth: int = 0, loops: bool = True) -> list:
    """A ZPAQL instruction sequence with matched if / do structures (libzpaq.cpp:6884-7190 compiles it)."""
    out = []
    for _ in range(rng.randrange(0, 12)):
        k = rng.randrange(12)
        if k == 0:
            out.append(rng.choice(_REG[:4]) + rng.choice(["++", "--", "!", "=0", "<>a"]) if rng.random() < .8 else "a=r %d" % rng.randrange(256))
        elif k == 1:
            out.append("%s=%s" % (rng.choice(_REG), rng.choice(_REG)))
        elif k == 2:
            out.append("%s= %d" % (rng.choice(_REG), rng.randrange(256)))
        elif k in (3, 4):
            out.append("a%s%s" % (rng.choice(_BIN), rng.choice(_REG)))
        elif k == 5:
            out.append("a%s %d" % (rng.choice(_BIN), rng.randrange(256)))
        elif k == 6:
            out.append(rng.choice(["hash", "hashd", "out", "r=a %d" % rng.randrange(256), "%s=r %d" % (rng.choice(_REG[:4]), rng.randrange(256))]))
        elif k == 7 and depth < 3:
            out += [rng.choice(["if", "ifnot", "ifl", "ifnotl"])] + random_code(rng, depth + 1, loops)
            if rng.random() < .5:
                out += [rng.choice(["else", "elsel"])] + random_code(rng, depth + 1, loops)
            out.append("endif")
        elif k == 8 and depth < 3 and loops:
            out += ["do"] + random_code(rng, depth + 1) + [rng.choice(["while", "until", "forever"])]
        elif k == 9:
            out.append("a+= $%d" % rng.randrange(1, 10) if rng.random() < .5 else "a= $%d+%d" % (rng.randrange(1, 10), rng.randrange(200)))
        elif k == 10 and rng.random() < .2:
            out.append(rng.choice(["jt 1", "jf 0", "jmp 2", "lj %d" % rng.randrange(40), "halt", "error"]))
    return out


def random_config(rng: random.Random) -> str:
    n = rng.randrange(0, 8)
    lines = ["comp %d %d %d %d %d" % (rng.randrange(0, 12), rng.randrange(0, 12), rng.randrange(0, 8), rng.randrange(0, 8), n)]
    for i in range(n):
        t = rng.choice(["const", "cm", "icm", "match", "avg", "mix2", "mix", "isse", "sse"]) if i else rng.choice(["const", "cm", "icm", "match"])
        j = rng.randrange(i) if i else 0
        k = rng.randrange(i) if i else 0
        sz = rng.randrange(0, 14)
        args = {"const": [rng.randrange(256)], "cm": [sz, rng.randrange(256)], "icm": [sz], "match": [sz, rng.randrange(0, 14)],
                "avg": [j, k, rng.randrange(256)], "mix2": [sz, j, k, rng.randrange(256), rng.randrange(256)],
                "mix": [min(sz, 10), j, rng.randrange(1, i - j + 1) if i else 1, rng.randrange(256), rng.randrange(256)], "isse": [sz, j],
                "sse": [sz, j, rng.randrange(0, 64), rng.randrange(64, 256)]}[t]
        lines.append("  %d %s %s" % (i, t, " ".join(map(str, args))))
    lines += ["hcomp"] + ["  " + " ".join(random_code(rng))] + ["  halt"]
    if rng.random() < .4:
        lines += ["pcomp some command ;"] + ["  " + " ".join(random_code(rng))] + ["  halt"]
    lines.append("end")
    return "\n".join(lines)


def seeds(z):
    from zpaq_amd import corpus
    text = bytes(corpus.zipf_text(6000, 3))
    noise = bytes(corpus.lcg_bytes(3000, 5))
    exe = b"".join(b"\xe8" + bytes(4) if i % 7 == 0 else bytes([i & 255]) for i in range(400))
    out = []
    for data in (text, noise, exe, b"", b"a"):
        for m in ("0", "1", "2", "x0,0", "x4,3", "x0,5,4,0,2", "x0,6,4,0,2", "x0,7"):
            try:
                out.append(z.compress_block(data, m, "name", "comment"))
            except z.ZpaqError:
                pass
    # two blocks back to back, and a block with leading garbage (the locator tag search)
    out.append(out[0] + out[1])
    out.append(b"junk" * 5 + out[2])
    return out


def _ref_decompress(archive: bytes, cap: int, timeout: float = 3.0):
    """The reference on a (possibly damaged) archive in a child process: bytes, "error", or "hang" (a damaged PCOMP
    program or BWT stream makes the reference loop for good)."""
    import multiprocessing as mp

    def child(q):
        try:
            from oracle.oracle_py import Ref
            q.put(Ref().decompress(archive, cap))
        except Exception:
            q.put("error")

    q = mp.Queue()
    pr = mp.Process(target=child, args=(q,))
    pr.start()
    try:
        r = q.get(timeout=timeout)
    except Exception:
        r = "hang" if pr.is_alive() else "crash"
    if pr.is_alive():
        pr.kill()
    pr.join()
    return r


def run(iterations: int, seed: int, verbose: bool = False, ref=None) -> dict:
    """ref: an oracle.oracle_py.Ref (the reference library built under oracle/_ref) -- every case is then run through the
    reference as well: what assembles must assemble to the same bytes, what is rejected must be rejected by both, what
    decodes must decode to the same bytes."""
    import zpaq_amd as z
    import ctypes as C
    rng = random.Random(seed)
    arch = seeds(z)
    z.lib().zpq_set_pcomp_step_limit.argtypes = [C.c_uint64]
    z.lib().zpq_set_pcomp_step_limit.restype = None
    z.lib().zpq_set_pcomp_step_limit(1 << 22)        # a damaged PCOMP program that loops gives up in milliseconds, not a minute
    diffs = []
    stats = {"decoded": 0, "rejected": 0, "plans": 0, "bad_plans": 0, "asm_ok": 0, "asm_bad": 0, "methods_ok": 0, "methods_bad": 0}
    cfg_words = ["comp", "hcomp", "pcomp", "end", "halt", "a=b", "b=a", "*c=a", "d++", "a<<=", "jt", "jf", "jmp", "lj", "if", "ifnot", "else",
                 "endif", "do", "while", "until", "forever", "icm", "isse", "cm", "mix", "mix2", "avg", "sse", "match", "const", "hash",
                 "hashd", "out", "a+=", "a==", "r=a", "a=r", "$1", "$2+3", "(", ")", "0", "1", "2", "3", "16", "255", "256", "-1", "99999", ";", "x"]
    t0 = time.time()
    trace = os.environ.get("FUZZ_TRACE")     # file that always holds the input being tried (for hangs and crashes)

    def note(kind, payload):
        if trace:
            with open(trace, "wb") as fh:
                fh.write(kind.encode() + b"\n" + (payload if isinstance(payload, bytes) else repr(payload).encode()))

    for it in range(iterations):
        which = it % 4
        if which == 0:
            a = mutate(rng, rng.choice(arch))
            if rng.random() < 0.3:
                a = mutate(rng, a)
            note("archive", a)
            try:
                mine = z.decompress(a, cap=1 << 20)
                stats["decoded"] += 1
            except z.ZpaqError as ex_:
                ex = ex_
                mine = "error"
                stats["rejected"] += 1
                if "too big" in str(ex):
                    mine = None                       # hh > 24 / hm > 28: this library's documented limit
                if "checksum mismatch" in str(ex):
                    mine = "checksum"                 # zpq_decompress verifies the SHA-1 trailers, libzpaq::decompress discards them
                if "[NODEVICE]" in str(ex) or "[DEVICE]" in str(ex):
                    mine = None                       # damaged into a context-mixing block: needs the GPU, not comparable here
            if ref is not None and mine is not None and it % 16 == 0:
                theirs = _ref_decompress(a, 1 << 20)
                stats["ref_" + (theirs if isinstance(theirs, str) else "decoded")] = stats.get("ref_" + (theirs if isinstance(theirs, str) else "decoded"), 0) + 1
                # a reference hang is the documented deviation (step bound); everything else must agree
                if mine == "checksum" and theirs != "hang":
                    stats["stricter_checksum"] = stats.get("stricter_checksum", 0) + 1
                elif theirs != "hang" and theirs != mine and mine == "error" and "[VM]" in str(ex):
                    # gave up on the small step budget of this run where the reference finished: again with a large one
                    z.lib().zpq_set_pcomp_step_limit(1 << 30)
                    try:
                        again = z.decompress(a, cap=1 << 20)
                    except z.ZpaqError as ex2:
                        again = "checksum" if "checksum mismatch" in str(ex2) else "error"
                    z.lib().zpq_set_pcomp_step_limit(1 << 22)
                    if again != theirs and again != "checksum":
                        diffs.append(("archive", a.hex(), repr(again)[:80], repr(theirs)[:80]))
                elif theirs != "hang" and theirs != mine:
                    diffs.append(("archive", a.hex(), repr(mine)[:80], repr(theirs)[:80]))
        elif which == 1:
            # block headers: a valid one damaged, or noise with a plausible length prefix
            if rng.random() < 0.7:
                src = rng.choice(arch)
                i = src.find(b"zPQ")
                h = bytearray(src[i + 5:i + 5 + 2 + src[i + 5] + 256 * src[i + 6]]) if i >= 0 else bytearray()
                h = bytearray(mutate(rng, bytes(h)))
            else:
                n = rng.randrange(0, 200)
                body = bytes(rng.randrange(256) for _ in range(n))
                h = bytearray([n & 255, n >> 8]) + body
            note("header", bytes(h))
            try:
                z.Plan(bytes(h))
                stats["plans"] += 1
            except z.ZpaqError:
                stats["bad_plans"] += 1
        elif which == 2:
            if rng.random() < 0.8:
                txt = random_config(rng)
                if rng.random() < 0.3:                # damage it: drop, double or swap a word
                    w = txt.split(" ")
                    i = rng.randrange(len(w))
                    w[i:i + 1] = rng.choice([[], [w[i], w[i]], [rng.choice(cfg_words)]])
                    txt = " ".join(w)
            else:
                txt = " ".join(rng.choice(cfg_words) for _ in range(rng.randrange(1, 60)))
            args = [rng.randrange(-5, 40) for _ in range(rng.randrange(0, 10))]
            note("config", (txt, args))
            theirs = None
            if ref is not None:
                try:
                    theirs = ref.compile(txt, args)
                except Exception:
                    theirs = "error"
            try:
                hdr, _pcomp = z.assemble(txt, args)
                if theirs is not None and theirs != (hdr, _pcomp):
                    diffs.append(("config", txt, args, "assembled differently" if theirs != "error" else "reference rejects"))
                stats["asm_ok"] += 1
                note("header", hdr)
                try:
                    z.Plan(hdr)                       # what assembles need not be a valid model (component ranges): error or plan
                    stats["plans"] += 1
                except z.ZpaqError:
                    stats["bad_plans"] += 1
            except z.ZpaqError:
                if theirs is not None and theirs != "error":
                    diffs.append(("config", txt, args, "reference accepts"))
                stats["asm_bad"] += 1
        else:
            parts = [rng.choice("0123456xs") + str(rng.randrange(0, 30))]
            for _ in range(rng.randrange(0, 8)):
                if rng.random() < 0.3:
                    parts.append(rng.choice("cimtawsf") + ".".join(str(rng.randrange(0, 1200)) for _ in range(rng.randrange(0, 5))))
                else:
                    parts.append(str(rng.randrange(0, 40)))
            m = ",".join(parts)
            note("method", m)
            x = None
            try:
                x = z.expand_method(m, bytes(rng.randrange(256) for _ in range(rng.randrange(0, 64))))
                mine = z.method_to_header(x)
                stats["methods_ok"] += 1
            except z.ZpaqError as ex:
                mine = "error"
                stats["methods_bad"] += 1
                if "Unsupported method" in str(ex) or "index-block" in str(ex):
                    x = None      # pre-processor type > 7 (the reference writes an archive nothing can restore) and 'i' methods: refused here
            if ref is not None and x is not None:
                try:
                    cfg, a9 = ref.make_config(x)
                    theirs = ref.compile(cfg, a9) + (a9,)
                except Exception:
                    theirs = "error"
                if theirs != mine:
                    diffs.append(("method", m, x, "reference: " + ("rejects" if theirs == "error" else "differs")))
        if verbose and it % 2000 == 1999:
            print(it + 1, "%.0f s" % (time.time() - t0), stats, flush=True)
    z.lib().zpq_set_pcomp_step_limit(0)
    if ref is not None:
        stats["differences"] = diffs
    return stats


if __name__ == "__main__":
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(pos[0]) if pos else 20000
    s = int(pos[1]) if len(pos) > 1 else 1
    r = None
    if "--ref" in sys.argv:
        from oracle.oracle_py import Ref
        r = Ref()
    st = run(n, s, verbose=True, ref=r)
    for d in st.pop("differences", []):
        print("DIFFERENCE", d)
    print(st)
