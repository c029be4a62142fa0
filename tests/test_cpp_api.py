"""The libzpaq-compatible C++ API (include/libzpaq.h + libzpaq_amd.so) used from a C++ program the
way zpaq.cpp uses libzpaq: compiled here with g++ against the in-tree library."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "compat_test")
    cmd = ["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "cpp", "compat_test.cpp"),
           "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "zpaq_amd"), "-lzpaq_amd",
           "-Wl,-rpath," + os.path.join(ROOT, "zpaq_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_cpp_api_host_paths(tmp_path, zlib_, golden):
    exe = _build(tmp_path)
    hdrs = [e["header"] for e in golden["level_cases"]]      # headers the reference wrote for startBlock(1|2|3)
    r = subprocess.run([exe, "host", *hdrs], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "COMPAT_OK" in r.stdout, r.stdout


def test_streaming_compress_of_several_blocks_host_methods(tmp_path, zlib_, ref):
    """libzpaq::compress() cuts its input into blocks (method "N0": 2^20 - 4096 bytes each), names only the first segment and
    batches the blocks; methods 0, 1, 2 have no model, so the whole thing runs without a GPU.  The archive must be the
    reference's byte for byte, and decompress() must return the input."""
    import numpy as np
    from zpaq_amd import corpus
    exe = _build(tmp_path)
    data = np.concatenate([corpus.block(k, n, 60 + i) for i, (k, n) in enumerate([("text", 1500000), ("records", 900000), ("lcg", 300000)])]).tobytes()
    src = tmp_path / "in.bin"
    src.write_bytes(data)
    for method in ("00", "10", "20", "11"):
        out = tmp_path / f"out{method}.zpaq"
        r = subprocess.run([exe, "stream", method, str(src), str(out), "file.bin", "a comment"], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0 and "COMPAT_OK" in r.stdout, r.stdout
        assert out.read_bytes() == ref.compress(data, method, "file.bin", "a comment"), method
    empty = tmp_path / "empty.bin"
    empty.write_bytes(b"")
    out = tmp_path / "empty.zpaq"
    r = subprocess.run([exe, "stream", "10", str(empty), str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    assert r.returncode == 0 and out.read_bytes() == ref.compress(b"", "10"), r.stdout


@pytest.mark.gpu
def test_cpp_api_on_gpu(tmp_path, gpu):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "gpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "COMPAT_OK" in r.stdout, r.stdout


@pytest.mark.gpu
def test_builtin_models_through_our_startblock(tmp_path, gpu, golden):
    """Compressor::startBlock(1|2|3) (min / mid / max.cfg) through OUR Compressor: the whole archive must be the
    one the reference produced for the same input (golden level_cases were made by the reference)."""
    import base64
    from conftest import gen_input
    exe = _build(tmp_path)
    for level, e in enumerate(golden["level_cases"], start=1):
        src, dst = str(tmp_path / f"in{level}"), str(tmp_path / f"out{level}")
        open(src, "wb").write(gen_input(e).tobytes())
        args = [exe, "level", str(level), src, dst, "lvl", "20000"]      # filename / comment tests/golden/make_golden.py used
        r = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0 and "COMPAT_OK" in r.stdout, r.stdout
        assert open(dst, "rb").read() == base64.b64decode(e["archive_b64"]), level


@pytest.mark.gpu
def test_concurrent_callers_are_coalesced(tmp_path, gpu):
    """32 threads, one libzpaq::compressBlock(.., "5") each, as zpaq.cpp's compressThread pool does: archives
    identical to the serial ones, and the whole thing takes about one batch, not 32 launches in a row."""
    exe = _build(tmp_path)
    r = subprocess.run([exe, "threads", "32", "5"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "COMPAT_OK" in r.stdout, r.stdout


@pytest.mark.gpu
def test_blocks_of_several_segments(tmp_path, gpu, ref):
    """Compressor: three segments in one modelled block (mid.cfg and max.cfg) -- the model and the coder run on from
    segment to segment (libzpaq.cpp:2889-2891, 2129).  Archive identical to the reference's; decoded both through
    libzpaq::decompress and segment by segment through Decompresser (inside the C++ program)."""
    from zpaq_amd import corpus
    exe = _build(tmp_path)
    parts = [corpus.block("text", 30000, 61).tobytes(), corpus.block("records", 17000, 62).tobytes(), b"", corpus.block("text", 9000, 63).tobytes()]
    paths = []
    for i, p in enumerate(parts):
        paths.append(str(tmp_path / f"part{i}"))
        open(paths[-1], "wb").write(p)
    for level in (2, 3):
        dst = str(tmp_path / f"multi{level}.zpaq")
        r = subprocess.run([exe, "segments", str(level), dst, *paths], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0 and "COMPAT_OK" in r.stdout, r.stdout
        ours = open(dst, "rb").read()
        assert ours == ref.compress_level_segments(parts, level), level
        assert gpu.decompress(ours) == b"".join(parts)


@pytest.mark.gpu
def test_decompress_n_decodes_a_prefix_on_the_device(tmp_path, gpu):
    """Decompresser::decompress(n) (libzpaq.cpp:2315-2343; zpaq.cpp:2859-2866 stops once it has the fragments it wants): the
    first 16 KiB of a 1 MiB -m5 block must not cost the whole block's decode.  tests/cpp/decomp_driver.cpp, mode 3: one
    decompress(16384) call, then readSegmentEnd -- the bytes are the input's first 16 KiB and the call takes a fraction of
    what decompress() to the end takes; read on in 16 KiB pieces (mode 0) the hand-over from the decoded prefix to the whole
    segment leaves no seam: every byte, the segment's SHA-1 and the trailer agree."""
    import hashlib
    import re
    import time
    from zpaq_amd import corpus
    drv = os.path.join(ROOT, "tests", "cpp", "decomp_driver.cpp")
    exe = str(tmp_path / "decomp_mine")
    r = subprocess.run(["g++", "-O1", "-std=c++17", drv, "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "zpaq_amd"), "-lzpaq_amd",
                        "-Wl,-rpath," + os.path.join(ROOT, "zpaq_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    data = corpus.block("text", 1 << 20, 77).tobytes()
    path = str(tmp_path / "one.zpaq")
    open(path, "wb").write(gpu.compress_blocks([data], "5")[0])
    small = corpus.block("text", 200_000, 78).tobytes()          # (pieces across the end of the 64 KiB prefix: a shorter block, one whole decode of 1 MiB is enough)
    path2 = str(tmp_path / "two.zpaq")
    open(path2, "wb").write(gpu.compress_blocks([small], "5")[0])

    def fnv(b):
        h = 1469598103934665603
        for c in b:
            h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return "%016x" % h

    def run(archive, piece, mode):
        p = subprocess.run([exe, archive, str(piece), str(mode)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert p.returncode == 0 and "error" not in p.stdout, (p.stdout[-400:], p.stderr[-400:])
        return p.stdout, p.stderr

    out, err = run(path2, 16384, 0)                         # (also the first process on this box: code object load, buffers)
    assert f"data n={len(small)} fnv={fnv(small)} sha_n={len(small)} sha={hashlib.sha1(small).hexdigest()}" in out, out
    out, err = run(path, 16384, 3)
    first_ms = float(re.search(r"first_call_ms=([0-9.]+)", err).group(1))
    assert f"data n=16384 fnv={fnv(data[:16384])}" in out, out
    t0 = time.time()
    out, err = run(path, -1, 0)                             # decompress() to the end: the whole block (it returns false at the end: the process is timed)
    whole_ms = (time.time() - t0) * 1e3
    assert f"data n={len(data)} fnv={fnv(data)}" in out, out
    assert first_ms < 0.35 * whole_ms, (first_ms, whole_ms)
    print(f"first 16 KiB: {first_ms:.0f} ms, whole block: {whole_ms:.0f} ms")


def _build_both(tmp_path, source, name):
    """One driver source against this library's libzpaq.h + .so, and against the reference's libzpaq.h + libzpaq.cpp."""
    ref_dir = "/root/reference"
    if not os.path.exists(os.path.join(ref_dir, "libzpaq.cpp")):
        pytest.skip("reference sources not present")
    drv = os.path.join(ROOT, "tests", "cpp", source)
    mine, theirs = str(tmp_path / (name + "_mine")), str(tmp_path / (name + "_ref"))
    for cmd in (["g++", "-O1", "-std=c++17", drv, "-I" + os.path.join(ROOT, "include"), "-L" + os.path.join(ROOT, "zpaq_amd"), "-lzpaq_amd",
                 "-Wl,-rpath," + os.path.join(ROOT, "zpaq_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", mine],
                ["g++", "-O1", "-std=c++17", "-DNDEBUG", "-Dunix", drv, "-I" + ref_dir, os.path.join(ref_dir, "libzpaq.cpp"), "-pthread", "-o", theirs]):
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    return mine, theirs


def test_compressor_call_sequences_write_the_reference_archives(tmp_path, zlib_):
    """tests/cpp/comp_driver.cpp through both libraries: Compressor on blocks without a model -- config text with and without
    a PCOMP section and its command, header bytes, postProcess() with and without a program, setVerify, segments with and
    without names / comments / checksums, compress() in pieces, endSegmentChecksum / getSize / getChecksum, calls out of
    order, a config error: same archive bytes, same return values, same error text."""
    mine, theirs = _build_both(tmp_path, "comp_driver.cpp", "comp")
    for scenario in range(7):
        for seed in range(1, 11):
            out = [subprocess.run([exe, str(scenario), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
                   for exe in (mine, theirs)]
            assert out[0].returncode == 0 and out[1].returncode == 0, (scenario, seed, out[0].stdout[-300:], out[1].stdout[-300:])
            assert out[0].stdout == out[1].stdout, (scenario, seed, out[0].stdout[:600], out[1].stdout[:600])
            assert "archive=" in out[0].stdout


def test_decompresser_call_sequences_match_the_reference_on_valid_and_damaged_archives(tmp_path, zlib_, ref):
    """tests/cpp/decomp_driver.cpp -- ONE source, built against include/libzpaq.h + libzpaq_amd.so and against the
    reference's libzpaq.h + libzpaq.cpp -- walks archives the way zpaq.cpp does (findBlock(&mem), hcomp, findFilename,
    readComment, decompress() / decompress(n) loops, pcomp, readSegmentEnd, skipped and half-read segments).  Blocks
    without a model (methods 0-2, BWT without a model) decode on the host, so this runs without a GPU.  The two programs
    must print the same events: memory figures, headers, names, comments, bytes, SHA-1 state and trailers, the position
    in the input (buffered()).  Damaged archives: both reject, or both print the same (what the reference delivered
    before it noticed is not compared, nor is how much of a half-read segment had left its output buffer)."""
    import random
    import fuzz_host
    mine, theirs = _build_both(tmp_path, "decomp_driver.cpp", "decomp")
    rng = random.Random(5)
    seeds = fuzz_host.seeds(zlib_)
    seeds.append(seeds[0] + seeds[3])           # several blocks in one stream
    # blocks of several segments that share one post-processor (custom PCOMP programs, unnamed segments, with and without
    # checksums): what tests/cpp/comp_driver.cpp writes through the Compressor
    cmine, _ = _build_both(tmp_path, "comp_driver.cpp", "comp")
    for scenario in range(4):
        for seed in (1, 2, 3):
            r = subprocess.run([cmine, str(scenario), str(seed)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
            arch = [l for l in r.stdout.splitlines() if l.startswith("archive=")]
            assert r.returncode == 0 and arch, r.stdout[-300:]
            seeds.append(bytes.fromhex(arch[0][len("archive="):]))
    path = str(tmp_path / "a.zpaq")

    def run(exe, piece, mode):
        try:
            p = subprocess.run([exe, path, str(piece), str(mode)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=3)
            return p.returncode, p.stdout.decode(), p.stderr.decode()
        except subprocess.TimeoutExpired:
            return "hang", "", ""

    same = errors = 0
    for it in range(3 * len(seeds) + 90):
        if it < 3 * len(seeds):
            a, mode = seeds[it % len(seeds)], it // len(seeds)      # every valid archive in every mode
        else:
            a, mode = fuzz_host.mutate(rng, rng.choice(seeds)), rng.choice([0, 0, 1, 2])
        piece = rng.choice([-1, -1, 1, 7, 100, 4096, 1 << 20])
        with open(path, "wb") as fh:
            fh.write(a)
        t = run(theirs, piece, mode)
        if t[0] == "hang":
            continue                     # a damaged PCOMP program: the reference never gives up
        m = run(mine, piece, mode)
        if "NODEVICE" in m[2] or "no GPU" in m[2] or "device" in m[2].lower():
            continue                     # damaged into a modelled block
        assert m[0] == 0 and t[0] == 0, (it, m, t)
        if mode == 2:
            m, t = [(o[0], "".join(l + "\n" for l in o[1].splitlines() if not l.startswith("data "))) for o in (m, t)]
        if m[1].rstrip().endswith("error") and t[1].rstrip().endswith("error"):
            errors += 1
            continue
        assert m[1] == t[1], (it, piece, mode, a.hex()[:400], m[1][-400:], t[1][-400:])
        same += 1
    assert same > 3 * len(seeds) and errors > 10, (same, errors)
