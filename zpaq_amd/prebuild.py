"""Ahead-of-time compilation of the per-header specialised kernels.

    python -m zpaq_amd.prebuild            (called by __graft_entry__.build())

For every standard block header -- the chains compressBlock generates for
methods "4" and "5" (.. "9") at each block-size exponent, with and without the
text hint, plus the headers recorded in tests/golden/golden.json (legacy
min/mid/max models, the all-nine-types config, the random HCOMP programs) -- ask the library for the
generated HIP source (zpq_plan_spec_source) and compile it with
`hipcc --genco --offload-arch=gfx950` into zpaq_amd/spec_cache/<key>.hsaco.
hipcc cross-compiles without a GPU.  Headers not covered here (e.g. level-5
chains with data-dependent periodic models) are compiled at run time by hipRTC
and added to the same cache.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def standard_headers():
    import zpaq_amd as z
    hs = {}
    for level in ("4", "5"):
        for arg0 in range(0, 7):
            for hint in ("", ",128,1"):        # plain / text
                # expand against a dummy buffer of the right size class: only n matters for arg0
                n = max(1, (1 << (20 + arg0)) - 4096)
                dummy = bytes(1)            # period detection sees no period in a 1-byte buffer
                xm = z.expand_method(level + hint, dummy)
                # expand_method derives arg0 from the buffer length: patch it in
                xm = "x" + str(arg0) + xm[xm.index(","):]
                try:
                    h, p, _ = z.method_to_header(xm)
                except z.ZpaqError:
                    continue
                hs[h] = f"method {level}{hint} arg0={arg0}"
    # BASELINE configs[1]: -m3 on 256 KiB blocks of LCG bytes (bench.py's `configs1` object)
    try:
        from zpaq_amd import corpus
        h, _, _ = z.method_to_header(z.expand_method("3", corpus.block("lcg", 1 << 18, corpus.BASE_SEED)))
        if h[6]:
            hs.setdefault(h, "method 3, configs[1]")
    except z.ZpaqError:
        pass
    gpath = os.path.join(ROOT, "tests", "golden", "golden.json")
    if os.path.exists(gpath):
        g = json.load(open(gpath))
        for sect in ("method_cases", "config_cases", "level_cases", "vm_cases"):
            for e in g[sect]:
                h = bytes.fromhex(e["header"])
                if h[6]:
                    hs.setdefault(h, f"golden {sect}")
    return hs


def source_and_key(header):
    import zpaq_amd as z
    L = z.lib()
    L.zpq_plan_spec_source.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p]
    p = z.Plan(header)
    buf = C.create_string_buffer(4 << 20)
    ln = C.c_size_t(0)
    key = C.create_string_buffer(41)
    rc = L.zpq_plan_spec_source(p._h, buf, len(buf), C.byref(ln), key)
    if rc != 0:
        return None, L.zpq_last_error().decode()
    return buf.value.decode(), key.value.decode()


def dual_source_and_key(header):
    """The decoder with two blocks per wavefront (device/spec_dual_kernel.h): source + cache key, or (None, reason)."""
    import zpaq_amd as z
    L = z.lib()
    L.zpq_plan_spec_dual_source.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p]
    p = z.Plan(header)
    buf = C.create_string_buffer(4 << 20)
    ln = C.c_size_t(0)
    key = C.create_string_buffer(41)
    rc = L.zpq_plan_spec_dual_source(p._h, buf, len(buf), C.byref(ln), key)
    if rc != 0:
        return None, L.zpq_last_error().decode()
    return buf.value.decode(), key.value.decode()


def team_source_and_key(header):
    """The lockstep decoder (device/spec_team_kernel.h): source + cache key, or (None, reason)."""
    import zpaq_amd as z
    L = z.lib()
    L.zpq_plan_spec_team_source.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p]
    p = z.Plan(header)
    buf = C.create_string_buffer(4 << 20)
    ln = C.c_size_t(0)
    key = C.create_string_buffer(41)
    rc = L.zpq_plan_spec_team_source(p._h, buf, len(buf), C.byref(ln), key)
    if rc != 0:
        return None, L.zpq_last_error().decode()
    return buf.value.decode(), key.value.decode()


def pipe_source_and_key(header, mode=0):
    """The pipelined encoder of this header (device/pipe_kernel.h) in one of its two shapes (mode 0 throughput, 1 latency):
    source + cache key, or (None, reason)."""
    import zpaq_amd as z
    L = z.lib()
    L.zpq_plan_pipe_source_opts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p]
    p = z.Plan(header)
    buf = C.create_string_buffer(4 << 20)
    ln = C.c_size_t(0)
    key = C.create_string_buffer(41)
    rc = L.zpq_plan_pipe_source_opts(p._h, mode, 0, 0, buf, len(buf), C.byref(ln), key)
    if rc != 0:
        return None, L.zpq_last_error().decode()
    return buf.value.decode(), key.value.decode()


def standard_pcomp_programs():
    """PCOMP programs of the pre-processing methods compressBlock generates (levels 1-4, every block type branch and
    block-size exponent): {(code, ph, pm): description}."""
    import zpaq_amd as z
    progs = {}
    for arg0 in range(0, 7):
        for body in (",1,4,0,3,%d" % (19 + arg0 + (arg0 <= 6)), ",5,4,0,3,%d" % (19 + arg0 + (arg0 <= 6)),
                     ",2,12,0,7,%d,1c0,0,511i2" % (21 + arg0), ",6,12,0,7,%d,1c0,0,511i2" % (21 + arg0),
                     ",2,5,0,7,%d1c0,0,511" % (21 + arg0), ",6,5,0,7,%d1c0,0,511" % (21 + arg0),
                     ",3ci1", ",7ci1", ",4ci1,1,1,1,2am", ",4"):
            xm = "x%d%s" % (arg0, body)
            try:
                h, pc, _ = z.method_to_header(xm)
            except z.ZpaqError:
                continue
            if pc:
                progs[(bytes(pc[2:]), h[4], h[5])] = xm
    return progs


def pcomp_source_and_key(code, ph, pm):
    import zpaq_amd as z
    L = z.lib()
    L.zpq_pcomp_source.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p]
    buf = C.create_string_buffer(4 << 20)
    ln = C.c_size_t(0)
    key = C.create_string_buffer(41)
    rc = L.zpq_pcomp_source(code, len(code), ph, pm, buf, len(buf), C.byref(ln), key)
    if rc != 0:
        return None, L.zpq_last_error().decode()
    return buf.value.decode(), key.value.decode()


def compile_one(args):
    src, key, cache, inc = args
    out = os.path.join(cache, key + ".hsaco")
    if os.path.exists(out) and os.path.getsize(out) > 0:
        return key, "cached"
    tmp = os.path.join(cache, key + ".hip")
    with open(tmp, "w") as fh:
        fh.write(src)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-label",
           "-mllvm", "-simplifycfg-sink-common=false",     # keeps the kernel's register state out of scratch
           "-I", inc, "--genco", tmp, "-o", out + ".tmp"]
    if os.environ.get("ZPAQ_AMD_SPEC_DEFS"):      # extra options separated by blanks (the cache key covers them: spec_loader.cpp)
        cmd[1:1] = os.environ["ZPAQ_AMD_SPEC_DEFS"].split()
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    os.remove(tmp)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {key}:\n{r.stdout[-3000:]}")
    os.replace(out + ".tmp", out)
    return key, "built"


# Code-generation knobs that tests keep bit-exact although they are off by default (built, measured on the MI355X, slower:
# DESIGN.md section 10): the generator reads them once per process, so each variant is generated and built by a child.
KNOB_VARIANTS = [({"ZPAQ_AMD_MIX_PACKED": "1"}, "pipe"), ({"ZPAQ_AMD_TEAM_TAIL": "1"}, "team")]


def build_variant(which):
    """(child process, knobs in the environment) the -m5 text chain's pipelined encoder (throughput shape) or lockstep
    decoder: prints 'KEY <cache key>' and builds the code object."""
    import zpaq_amd as z
    from zpaq_amd import corpus
    L = z.lib()
    L.zpq_spec_cache_dir.restype = C.c_char_p
    L.zpq_spec_include_dir.restype = C.c_char_p
    cache, inc = L.zpq_spec_cache_dir().decode(), L.zpq_spec_include_dir().decode()
    os.makedirs(cache, exist_ok=True)
    h = z.method_to_header(z.expand_method("5", corpus.block("text", 1 << 20, corpus.BASE_SEED)))[0]
    src, key = pipe_source_and_key(h, 0) if which == "pipe" else team_source_and_key(h)
    if src is None:
        raise RuntimeError(key)
    print("KEY", key, flush=True)
    compile_one((src, key, cache, inc))


def knob_variants():
    """Builds KNOB_VARIANTS (children, side by side); returns their cache keys."""
    procs = [subprocess.Popen([sys.executable, "-m", "zpaq_amd.prebuild", "--variant", which], cwd=ROOT, env=dict(os.environ, **env),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for env, which in KNOB_VARIANTS]
    keys = []
    for pr, (env, which) in zip(procs, KNOB_VARIANTS):
        out = pr.communicate()[0]
        if pr.returncode != 0:
            raise RuntimeError(f"prebuild of variant {env} failed:\n{out[-3000:]}")
        keys += [l.split()[1] for l in out.splitlines() if l.startswith("KEY ")]
    return keys


def main(verbose=True, clean=True):
    import zpaq_amd as z
    L = z.lib()
    L.zpq_spec_cache_dir.restype = C.c_char_p
    L.zpq_spec_include_dir.restype = C.c_char_p
    cache = L.zpq_spec_cache_dir().decode()
    inc = L.zpq_spec_include_dir().decode()
    os.makedirs(cache, exist_ok=True)
    jobs, skipped = [], 0
    seen = set()
    # every header in both workgroup shapes (zpq_plan_spec_source reads ZPAQ_AMD_SPEC_WAVES): 4 blocks per
    # workgroup for batches of up to 4 x CUs blocks, 8 per workgroup (two wavefronts per SIMD) beyond that
    forced = os.environ.get("ZPAQ_AMD_SPEC_WAVES")
    for h, why in standard_headers().items():
        for mode in (0, 1, 2, 3):     # throughput, latency, latency with 2048-byte steps, latency with a wavefront per SIMD (host/codegen.hpp pipe_options)
            src, key = pipe_source_and_key(h, mode)
            if src is not None and key not in seen:
                seen.add(key)
                jobs.append((src, key, cache, inc))
        for variant in (dual_source_and_key, team_source_and_key):
            src, key = variant(h)
            if src is not None and key not in seen:
                seen.add(key)
                jobs.append((src, key, cache, inc))
        for waves in ("4", "8"):
            os.environ["ZPAQ_AMD_SPEC_WAVES"] = waves
            src, key = source_and_key(h)
            if src is None:
                skipped += waves == "4"
                continue
            if key in seen:
                continue
            seen.add(key)
            jobs.append((src, key, cache, inc))
    for (code, ph, pm), why in standard_pcomp_programs().items():
        src, key = pcomp_source_and_key(code, ph, pm)
        if src is not None and key not in seen:
            seen.add(key)
            jobs.append((src, key, cache, inc))
    if forced is None:
        del os.environ["ZPAQ_AMD_SPEC_WAVES"]
    else:
        os.environ["ZPAQ_AMD_SPEC_WAVES"] = forced
    seen.update(knob_variants())
    # drop stale code objects of older template versions
    for fn in os.listdir(cache):
        if clean and fn.endswith(".hsaco") and fn[:-6] not in seen:       # (--keep: experiments keep their variants)
            os.remove(os.path.join(cache, fn))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(compile_one, jobs))
    if verbose:
        print(f"spec_cache: {sum(1 for _, s in res if s == 'built')} built, "
              f"{sum(1 for _, s in res if s == 'cached')} up to date, {skipped} not specialisable -> {cache}")
    return res


if __name__ == "__main__":
    if "--variant" in sys.argv:
        build_variant(sys.argv[sys.argv.index("--variant") + 1])
    else:
        main(clean="--keep" not in sys.argv)
