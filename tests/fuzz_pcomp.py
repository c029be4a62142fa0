#!/usr/bin/env python3
"""Random PCOMP post-processing programs three ways: (1) the reference's PostProcessor (oracle/_ref, its interpreter,
decompressing a stored block that carries the program), (2) this library's host interpreter (zpq_decompress of the same archive),
(3) the TRANSLATOR that serves the device (zpq_pcomp_source: ZPAQL -> straight-line HIP C++), whose output is compiled
for the host against host/pcomp_host.h -- the way the standard programs are built into the library -- and run on the
same bytes.  All three must write the same output (or all must fail).  No GPU.

    python tests/fuzz_pcomp.py [programs] [seed]
"""
from __future__ import annotations

import ctypes as C
import os
import random
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "pcomp_host.h"
%s
int main(int argc, char** argv) {
  std::vector<uint8_t> in, out;
  FILE* f = fopen(argv[1], "rb");
  int c;
  while ((c = fgetc(f)) != EOF) in.push_back((uint8_t)c);
  fclose(f);
  const int ph = atoi(argv[3]), pm = atoi(argv[4]);
  std::vector<uint32_t> H((size_t)1 << ph, 0);
  std::vector<uint8_t> M((size_t)1 << pm, 0);
  zpq::PcompHostState s;
  s.H = H.data();
  s.M = M.data();
  const int st = zpq::pcomp_host_run<zpq_gen::Post>(s, in.data(), in.size(), true, out);
  f = fopen(argv[2], "wb");
  fwrite(out.data(), 1, out.size(), f);
  fclose(f);
  printf("status %%d\n", st);
  return 0;
}
'''


def random_program(rng: random.Random) -> str:
    import fuzz_host
    code = []
    for _ in range(rng.randrange(1, 4)):
        code += fuzz_host.random_code(rng, 0, False)
        if rng.random() < 0.5:          # a bounded loop: d counts down
            code += ["d= %d" % rng.randrange(1, 40), "do"] + [w for w in fuzz_host.random_code(rng, 1, False) if not w.startswith(("d", "*d=", "a<>d"))] + \
                    ["d--", "a=d", "a> 0", "while"]
    code = [w for w in code if not w.startswith(("error", "lj", "jt", "jf", "jmp", "halt", "a+= $", "a= $"))]
    # most programs should write something
    for _ in range(rng.randrange(1, 4)):
        code.insert(rng.randrange(len(code) + 1), "out")
    return " ".join(code)


def run(programs: int, seed: int, verbose: bool = True) -> int:
    import zpaq_amd as z
    from oracle.oracle_py import Ref
    from zpaq_amd import corpus
    L = z.lib()
    L.zpq_pcomp_source.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p]
    L.zpq_set_pcomp_step_limit.argtypes = [C.c_uint64]
    L.zpq_set_pcomp_step_limit.restype = None
    ref = Ref()
    # The yardstick is the reference's INTERPRETER (-DNOJIT), i.e. the ZPAQ specification: its x86 JIT executes two swaps in a
    # row ("b<>a b<>a") as one -- found by this fuzzer; programs on which the two disagree are counted, not compared.
    ref_interp = Ref(nojit=True)
    jit_differs = 0
    rng = random.Random(seed)
    inc = os.path.join(ROOT, "zpaq_amd", "csrc", "host")
    done = 0
    t0 = time.time()
    with tempfile.TemporaryDirectory() as td:
        while done < programs:
            ph, pm = rng.randrange(0, 6), rng.randrange(0, 10)
            cfg = "comp 0 0 %d %d 0 hcomp halt pcomp prog ; %s halt end" % (ph, pm, random_program(rng))
            try:
                header, pcomp = z.assemble(cfg)
            except z.ZpaqError:
                continue
            data = rng.choice([corpus.block("text", 300, rng.randrange(99)).tobytes(), bytes(rng.randrange(256) for _ in range(200)), bytes(100), b"", b"\xff" * 50])
            try:
                arch = ref.compress_config(data, cfg, None, "f", None, False)     # stores the bytes, writes the program in front
            except Exception:
                continue
            try:
                want = ref_interp.decompress(arch, 1 << 20)
            except Exception:
                want = None
            try:
                jit = ref.decompress(arch, 1 << 20)
            except Exception:
                jit = None
            jit_differs += jit != want
            try:
                mine = z.decompress(arch, cap=1 << 20)
            except z.ZpaqError:
                mine = None
            if mine != want:
                print("INTERPRETER DIFFERS", None if mine is None else len(mine), None if want is None else len(want), "\n" + cfg, flush=True)
                return 1
            # the translator
            code = pcomp[2:]
            buf = C.create_string_buffer(4 << 20)
            ln = C.c_size_t(0)
            key = C.create_string_buffer(41)
            rc = L.zpq_pcomp_source(code, len(code), ph, pm, buf, len(buf), C.byref(ln), key)
            if rc != 0:
                print("NOT TRANSLATED", L.zpq_last_error().decode(), "\n" + cfg, flush=True)
                return 1
            src = buf.value.decode()
            b, e = src.find("namespace zpq_gen {"), src.find("}  // namespace zpq_gen")
            assert b >= 0 and e >= 0
            cpp = os.path.join(td, "p.cpp")
            with open(cpp, "w") as fh:
                fh.write(DRIVER % (src[b:e] + "}\n"))
            exe = os.path.join(td, "p")
            r = subprocess.run(["g++", "-O1", "-std=c++17", "-w", "-I", inc, cpp, "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                print("TRANSLATION DOES NOT COMPILE\n" + r.stdout[-2000:] + "\n" + cfg, flush=True)
                return 1
            with open(os.path.join(td, "in"), "wb") as fh:
                fh.write(data)
            r = subprocess.run([exe, os.path.join(td, "in"), os.path.join(td, "out"), str(ph), str(pm)], stdout=subprocess.PIPE, text=True, timeout=120)
            got = open(os.path.join(td, "out"), "rb").read()
            status = int(r.stdout.split()[-1]) if r.returncode == 0 and r.stdout.startswith("status") else -999
            if (want is None) != (status != 0) or (want is not None and got != want):
                print("TRANSLATION DIFFERS status", status, len(got), None if want is None else len(want), "\n" + cfg, flush=True)
                return 1
            done += 1
            if verbose and done % 20 == 0:
                print("%d programs ok (%.0f s; reference JIT != reference interpreter on %d)" % (done, time.time() - t0, jit_differs), flush=True)
    return 0


if __name__ == "__main__":
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    sys.exit(run(int(pos[0]) if pos else 40, int(pos[1]) if len(pos) > 1 else 1))
