"""The libzpaq-compatible C++ API (include/libzpaq.h + libzpaq_amd.so) used from a C++ program the
way zpaq.cpp uses libzpaq: compiled here with g++ against the in-tree library."""
import os
import subprocess

import pytest

from conftest import ROOT


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
