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


def test_cpp_api_host_paths(tmp_path, zlib_):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "host"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "COMPAT_OK" in r.stdout, r.stdout


@pytest.mark.gpu
def test_cpp_api_on_gpu(tmp_path, gpu):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "gpu"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "COMPAT_OK" in r.stdout, r.stdout
