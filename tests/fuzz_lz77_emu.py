#!/usr/bin/env python3
"""Random LZ77 parameters and inputs through the device's parse on the host (tests/emu/lz77_emu_main.cpp runs
zpaq_amd/csrc/device/lz77_kernel.h lane by lane) against the host's parse of the same block: level 1 / 2 codes, minimum match
1..24, look-ahead 0..8, 1..128 candidates per direction, E8E9 in front, text / random / zeros / records / patterns / long
repeats / x86-like bytes, lengths from 0 to a few thousand (a few larger ones).  The two token lists must be identical; the
coded stream of the list must be what zpq_preprocess_block makes.  No GPU.

    python tests/fuzz_lz77_emu.py [rounds] [seed]"""
from __future__ import annotations

import ctypes as C
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "emu"))


def main():
    import emu
    import zpaq_amd as z
    from zpaq_amd import corpus
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    L = z.lib()
    u8p = C.POINTER(C.c_ubyte)
    u32p = C.POINTER(C.c_uint32)
    L.zpq_lz77_tokens_host.argtypes = [C.c_char_p, u8p, C.c_uint32, u32p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zpq_lz77_serialize.argtypes = [C.c_char_p, u8p, C.c_uint32, u32p, C.c_size_t, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.zpq_preprocess_block.argtypes = [C.c_char_p, u8p, C.c_uint32, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    kinds = ["text", "lcg", "zeros", "records", "pattern"]

    def data(n):
        k = rng.randrange(8)
        if k < 5:
            return corpus.block(kinds[k], n, rng.randrange(1 << 30)).tobytes() if n else b""
        if k == 5:                                  # long repeats with noise between them
            unit = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 40)))
            out = b""
            while len(out) < n:
                out += unit * rng.randrange(1, 400) + bytes(rng.randrange(256) for _ in range(rng.randrange(0, 30)))
            return out[:n]
        if k == 6:                                  # x86-like: E8 / E9 opcodes with small displacements
            a = bytearray(corpus.block("lcg", n, rng.randrange(1 << 30)).tobytes())
            for _ in range(n // 9):
                i = rng.randrange(max(1, n - 5))
                if i + 5 <= n:
                    a[i] = rng.choice([0xE8, 0xE9]); a[i + 4] = rng.choice([0, 255])
            return bytes(a)
        return bytes(rng.choice(b"ab") for _ in range(n))       # two symbols: many equally long candidates

    total = 0
    for r in range(rounds):
        level = rng.choice([1, 2])
        e8 = rng.random() < 0.3
        mm = rng.randrange(4, 13) if level == 1 else rng.randrange(1, 25)
        la = rng.choice([0, 0, 1, 1, 2, 3, 5, 8])
        lb = rng.randrange(0, 8)
        xm = "x0,%d,%d,0,%d,21,%d" % (level + (4 if e8 else 0), mm, lb, la) + (",c0,0,511" if level == 2 else "")
        a = z.method_to_header(xm)[2]
        ns = [rng.choice([0, 1, 2, 3, 5, 17, 64, 255, 256, 300, 1000, 2500, 4095, 4096, 4097, 6000, 9000]) for _ in range(6)]
        if r % 8 == 0:
            ns.append(rng.randrange(20000, 60000))
        ins = [data(n) for n in ns]
        host, bufs = [], []
        for d in ins:
            buf = np.frombuffer(bytearray(d), np.uint8).copy() if d else np.zeros(1, np.uint8)
            toks = np.zeros(4 * (len(d) + 4), np.uint32)
            cnt = C.c_size_t(0)
            assert L.zpq_lz77_tokens_host(xm.encode(), buf.ctypes.data_as(u8p), len(d), toks.ctypes.data_as(u32p), len(d) + 4, C.byref(cnt)) == 0, L.zpq_last_error()
            host.append(toks[:4 * cnt.value].copy())
            bufs.append(buf[:len(d)].tobytes())          # (E8E9 applied)
        got = emu.lz77_run(a[1] & 3, a[2], a[6], (1 << a[4]) - 1, 17 + a[0], bufs)
        for k, (h, g) in enumerate(zip(host, got)):
            assert h.tobytes() == g, (xm, k, ns[k], h.size // 4, len(g) // 16)
            # the coder: the list's stream = the pre-processor's stream
            d = ins[k]
            src = np.frombuffer(bytearray(d), np.uint8).copy() if d else np.zeros(1, np.uint8)
            want = np.empty(len(d) * 2 + 4096, np.uint8)
            wl = C.c_size_t(0)
            assert L.zpq_preprocess_block(xm.encode(), src.ctypes.data_as(u8p), len(d), want.ctypes.data_as(u8p), want.size, C.byref(wl)) == 0
            e = np.frombuffer(bytearray(bufs[k]), np.uint8).copy() if d else np.zeros(1, np.uint8)
            out = np.empty(want.size, np.uint8)
            ol = C.c_size_t(0)
            hh = h.copy() if h.size else np.zeros(4, np.uint32)
            assert L.zpq_lz77_serialize(xm.encode(), e.ctypes.data_as(u8p), len(d), hh.ctypes.data_as(u32p), h.size // 4, out.ctypes.data_as(u8p), out.size, C.byref(ol)) == 0, L.zpq_last_error()
            assert ol.value == wl.value and (out[:ol.value] == want[:wl.value]).all(), (xm, k, ns[k])
            total += 1
        print("round %d ok: %s, %d blocks" % (r + 1, xm, len(ins)), flush=True)
    print("blocks", total, "no mismatch")


if __name__ == "__main__":
    main()
