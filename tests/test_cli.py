"""End to end through the reference's own caller: oracle/_ref/zpaq_amd_cli is the UNMODIFIED reference zpaq.cpp built
against include/libzpaq.h + libzpaq_amd.so (oracle/Makefile `cli`); oracle/_ref/zpaq_ref_cli is the reference as it is.
`zpaq add` runs compressBlock from a pool of threads (zpaq.cpp:1918-1965) -- here they are coalesced by the library's
submission queue -- and `zpaq extract` runs Decompresser from a pool of threads (2848-2867)."""
import filecmp
import os
import subprocess

import pytest

from conftest import ROOT
from zpaq_amd import corpus

OURS = os.path.join(ROOT, "oracle", "_ref", "zpaq_amd_cli")
REF = os.path.join(ROOT, "oracle", "_ref", "zpaq_ref_cli")


def _tree(root):
    os.makedirs(os.path.join(root, "sub"))
    open(os.path.join(root, "a.txt"), "wb").write(corpus.block("text", 300000, 1).tobytes())
    open(os.path.join(root, "b.bin"), "wb").write(corpus.block("lcg", 150000, 2).tobytes())
    open(os.path.join(root, "sub", "c.rec"), "wb").write(corpus.block("records", 160000, 3).tobytes())
    open(os.path.join(root, "sub", "d.txt"), "wb").write(corpus.block("text", 40000, 4).tobytes() * 3)
    open(os.path.join(root, "sub", "empty"), "wb").write(b"")


def _same(a, b):
    c = filecmp.dircmp(a, b)
    if c.left_only or c.right_only or c.funny_files:
        return False
    _, mismatch, errors = filecmp.cmpfiles(a, b, c.common_files, shallow=False)
    return not mismatch and not errors and all(_same(os.path.join(a, d), os.path.join(b, d)) for d in c.common_dirs)


def _roundtrip(tmp_path, methods):
    if not (os.path.exists(OURS) and os.path.exists(REF)):
        pytest.skip("oracle/_ref CLIs not built (no /root/reference here)")
    src = str(tmp_path / "src")
    _tree(src)
    for m in methods:
        ours, ref = str(tmp_path / f"ours{m}.zpaq"), str(tmp_path / f"ref{m}.zpaq")
        for exe, arc in ((OURS, ours), (REF, ref)):
            r = subprocess.run([exe, "add", arc, "src", "-method", m, "-threads", "8"], cwd=str(tmp_path), capture_output=True, text=True, timeout=1200)
            assert r.returncode == 0, r.stderr[-2000:]
        assert os.path.getsize(ours) == os.path.getsize(ref), m       # (the bytes differ only in the timestamps of the index)
        for exe, arc, to in ((REF, ours, f"ref_from_ours{m}"), (OURS, ref, f"ours_from_ref{m}")):
            out = str(tmp_path / to)
            r = subprocess.run([exe, "extract", arc, "-to", out, "-threads", "8"], cwd=str(tmp_path), capture_output=True, text=True, timeout=1200)
            assert r.returncode == 0, r.stderr[-2000:]
            assert _same(src, os.path.join(out, "src")), (m, to)


BATCH = os.path.join(ROOT, "oracle", "_ref", "zpaq_amd_cli_batch")


def _batch_roundtrip(tmp_path, methods, files=40):
    """patches/zpaq_batch.patch: `add` hands CompressJob's whole queue to libzpaq::compressBlocks, `extract` decodes every
    READY block with one libzpaq::decompress call.  Same archive size as the reference's, and each extracts the other's."""
    if not (os.path.exists(BATCH) and os.path.exists(REF)):
        pytest.skip("oracle/_ref CLIs not built (no /root/reference here)")
    src = str(tmp_path / "src")
    os.makedirs(src)
    for i in range(files):
        open(os.path.join(src, f"f{i:03d}.bin"), "wb").write(corpus.block(["text", "lcg", "records", "zeros"][i % 4], 90000 + 7001 * i, 7 + i).tobytes())
    open(os.path.join(src, "empty"), "wb").write(b"")
    for m in methods:
        ours, ref = str(tmp_path / f"batch{m}.zpaq"), str(tmp_path / f"ref{m}.zpaq")
        for exe, arc in ((BATCH, ours), (REF, ref)):
            r = subprocess.run([exe, "add", arc, "src", "-method", m, "-threads", "4"], cwd=str(tmp_path), capture_output=True, text=True, timeout=1800)
            assert r.returncode == 0, r.stderr[-2000:]
        assert os.path.getsize(ours) == os.path.getsize(ref), m
        for exe, arc, to in ((REF, ours, f"ref_from_batch{m}"), (BATCH, ref, f"batch_from_ref{m}")):
            out = str(tmp_path / to)
            r = subprocess.run([exe, "extract", arc, "-to", out, "-threads", "4"], cwd=str(tmp_path), capture_output=True, text=True, timeout=1800)
            assert r.returncode == 0, r.stderr[-2000:]
            assert _same(src, os.path.join(out, "src")), (m, to)


def test_patched_archiver_hands_its_queues_to_the_batch_api_host_methods(tmp_path, zlib_):
    _batch_roundtrip(tmp_path, ["10", "2"])


@pytest.mark.gpu
def test_patched_archiver_hands_its_queues_to_the_batch_api_modelled_methods(tmp_path, gpu):
    _batch_roundtrip(tmp_path, ["30", "50"], files=24)


def test_reference_archiver_on_this_library_host_methods(tmp_path, zlib_):
    """-m0, -m1, -m2 have no context model: the whole path runs on the host (LZ77 + framing)."""
    _roundtrip(tmp_path, ["0", "1", "2"])


@pytest.mark.gpu
def test_reference_archiver_on_this_library_modelled_methods(tmp_path, gpu):
    """-m3 (LZ77 / BWT + model), -m4, -m5 (context mixing): zpaq.cpp's compressThread / decompressThread pools on the GPU."""
    _roundtrip(tmp_path, ["3", "4", "5"])
