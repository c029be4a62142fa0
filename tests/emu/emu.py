"""TEST INFRASTRUCTURE: build and run the host-side wavefront emulator (wave_emu.h) for one block header.

The kernel text is exactly what the engine would hand to hipcc / hipRTC (zpq_plan_spec_source); here it is
compiled for the host against wave_emu.h and executed lane by lane.  Used by tests/test_emu.py."""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess
import tempfile
from typing import List, Sequence

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EMU = os.path.join(ROOT, "tests", "emu")
BUILD = os.path.join(ROOT, "build", "emu")


def _sanitize_flags():
    """ZPQ_EMU_SANITIZE=1: the device code under UBSan: indices into the fixed arrays (LDS tables, per-slot registers),
    misaligned and null accesses, bad bool loads abort the run.  Not checked: shift counts and signed overflow -- idle
    lanes compute on their dummy parameters and discard the result (spec_kernel.h:515: a CONST lane's "sizebits" as a shift
    count; :199: the multiply-add of a lane that is no ISSE), and the hardware wraps both.  No AddressSanitizer: the
    lanes are fibers that switch stacks."""
    if os.environ.get("ZPQ_EMU_SANITIZE") != "1":
        return ()
    return ("-fsanitize=undefined", "-fno-sanitize=shift,signed-integer-overflow", "-fno-sanitize-recover=undefined", "-g")


def kernel_source(header: bytes, waves: int) -> str:
    import zpaq_amd as z
    L = z.lib()
    L.zpq_plan_spec_source.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p]
    plan = z.Plan(header)
    buf = C.create_string_buffer(4 << 20)
    ln = C.c_size_t(0)
    key = C.create_string_buffer(41)
    old = {k: os.environ.get(k) for k in ("ZPAQ_AMD_SPEC_WAVES",)}
    os.environ["ZPAQ_AMD_SPEC_WAVES"] = str(waves)
    try:
        rc = L.zpq_plan_spec_source(plan._h, buf, len(buf), C.byref(ln), key)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    if rc != 0:
        raise RuntimeError(L.zpq_last_error().decode())
    return buf.value.decode()


def build(header: bytes, waves: int, extra_flags: Sequence[str] = ()) -> str:
    """Compile the emulator executable for this header/shape (cached under build/emu); returns its path."""
    import zpaq_amd as z
    src = kernel_source(header, waves)
    deps = b"".join(open(p, "rb").read() for p in (
        os.path.join(EMU, "wave_emu.h"), os.path.join(EMU, "wave_emu.cpp"), os.path.join(EMU, "emu_main.cpp"), os.path.join(EMU, "guard_alloc.h"),
        os.path.join(ROOT, "zpaq_amd", "csrc", "device", "spec_kernel.h"),
        os.path.join(ROOT, "zpaq_amd", "csrc", "device", "layout.h")))
    extra_flags = tuple(extra_flags) + _sanitize_flags()
    key = hashlib.sha1(src.encode() + deps + " ".join(extra_flags).encode()).hexdigest()[:20]
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, f"emu_{key}")
    if os.path.exists(exe):
        return exe
    gen = os.path.join(BUILD, f"gen_{key}.cpp")
    with open(gen, "w") as fh:
        fh.write('#include "wave_emu.h"\n' + src)
    libdir = os.path.dirname(z.library_path())
    cmd = ["g++", "-O1", "-std=c++17", "-w", *extra_flags, "-I", EMU, "-I", os.path.join(ROOT, "zpaq_amd", "csrc", "device"),
           "-I", os.path.join(ROOT, "include"), gen, os.path.join(EMU, "emu_main.cpp"), os.path.join(EMU, "wave_emu.cpp"),
           "-L", libdir, "-lzpaq_amd", f"-Wl,-rpath,{libdir}", "-o", exe + ".tmp"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    os.remove(gen)
    if r.returncode != 0:
        raise RuntimeError("emulator build failed:\n" + r.stdout[-4000:])
    os.replace(exe + ".tmp", exe)
    return exe


def dual_source(header: bytes) -> str:
    """The translation unit of the two-blocks-per-wavefront decoder (zpq_plan_spec_dual_source)."""
    import zpaq_amd as z
    L = z.lib()
    L.zpq_plan_spec_dual_source.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p]
    plan = z.Plan(header)
    buf = C.create_string_buffer(4 << 20)
    ln = C.c_size_t(0)
    key = C.create_string_buffer(41)
    if L.zpq_plan_spec_dual_source(plan._h, buf, len(buf), C.byref(ln), key) != 0:
        raise RuntimeError(L.zpq_last_error().decode())
    return buf.value.decode()


def build_dual(header: bytes) -> str:
    import zpaq_amd as z
    src = dual_source(header)
    dev = os.path.join(ROOT, "zpaq_amd", "csrc", "device")
    deps = b"".join(open(p, "rb").read() for p in (
        os.path.join(EMU, "wave_emu.h"), os.path.join(EMU, "wave_emu.cpp"), os.path.join(EMU, "emu_main.cpp"), os.path.join(EMU, "guard_alloc.h"),
        os.path.join(dev, "spec_kernel.h"), os.path.join(dev, "spec_dual_kernel.h"), os.path.join(dev, "layout.h")))
    flags = _sanitize_flags()
    key = hashlib.sha1(src.encode() + deps + " ".join(flags).encode()).hexdigest()[:20]
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, f"dual_{key}")
    if os.path.exists(exe):
        return exe
    gen = os.path.join(BUILD, f"dgen_{key}.cpp")
    # the emulator's main also refers to the one-block entry points: stubs
    stubs = ('extern "C" void zpq_spec_encode(const zpq::BlockJob*, zpq::BlockResult*, unsigned, const zpq::DeviceTables*) {}\n'
             'extern "C" void zpq_spec_decode(const zpq::BlockJob*, zpq::BlockResult*, unsigned, const zpq::DeviceTables*) {}\n')
    with open(gen, "w") as fh:
        fh.write('#include "wave_emu.h"\n' + src + stubs)
    libdir = os.path.dirname(z.library_path())
    cmd = ["g++", "-O1", "-std=c++17", "-w", "-DZPQ_EMU_DUAL", *flags, "-I", EMU, "-I", dev, "-I", os.path.join(ROOT, "include"), gen,
           os.path.join(EMU, "emu_main.cpp"), os.path.join(EMU, "wave_emu.cpp"), "-L", libdir, "-lzpaq_amd", f"-Wl,-rpath,{libdir}", "-o", exe + ".tmp"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    os.remove(gen)
    if r.returncode != 0:
        raise RuntimeError("dual emulator build failed:\n" + r.stdout[-6000:])
    os.replace(exe + ".tmp", exe)
    return exe


def team_source(header: bytes) -> str:
    """The translation unit of the lockstep decoder (zpq_plan_spec_team_source)."""
    import zpaq_amd as z
    L = z.lib()
    L.zpq_plan_spec_team_source.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p]
    plan = z.Plan(header)
    buf = C.create_string_buffer(4 << 20)
    ln = C.c_size_t(0)
    key = C.create_string_buffer(41)
    if L.zpq_plan_spec_team_source(plan._h, buf, len(buf), C.byref(ln), key) != 0:
        raise RuntimeError(L.zpq_last_error().decode())
    return buf.value.decode()


def team_threads(src: str) -> int:
    import re
    return int(re.search(r"__launch_bounds__\((\d+)\) void zpq_spec_decode3", src).group(1))


def build_team(header: bytes):
    import zpaq_amd as z
    src = team_source(header)
    dev = os.path.join(ROOT, "zpaq_amd", "csrc", "device")
    deps = b"".join(open(p, "rb").read() for p in (
        os.path.join(EMU, "wave_emu.h"), os.path.join(EMU, "wave_emu.cpp"), os.path.join(EMU, "emu_main.cpp"), os.path.join(EMU, "guard_alloc.h"),
        os.path.join(dev, "spec_kernel.h"), os.path.join(dev, "spec_dual_kernel.h"), os.path.join(dev, "spec_team_kernel.h"),
        os.path.join(dev, "layout.h")))
    flags = _sanitize_flags()
    key = hashlib.sha1(src.encode() + deps + " ".join(flags).encode()).hexdigest()[:20]
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, f"team_{key}")
    if os.path.exists(exe):
        return exe, team_threads(src)
    gen = os.path.join(BUILD, f"tgen_{key}.cpp")
    stubs = ('extern "C" void zpq_spec_encode(const zpq::BlockJob*, zpq::BlockResult*, unsigned, const zpq::DeviceTables*) {}\n'
             'extern "C" void zpq_spec_decode(const zpq::BlockJob*, zpq::BlockResult*, unsigned, const zpq::DeviceTables*) {}\n')
    with open(gen, "w") as fh:
        fh.write('#include "wave_emu.h"\n' + src + stubs)
    libdir = os.path.dirname(z.library_path())
    cmd = ["g++", "-O1", "-std=c++17", "-w", "-DZPQ_EMU_TEAM", *flags, "-I", EMU, "-I", dev, "-I", os.path.join(ROOT, "include"), gen,
           os.path.join(EMU, "emu_main.cpp"), os.path.join(EMU, "wave_emu.cpp"), "-L", libdir, "-lzpaq_amd", f"-Wl,-rpath,{libdir}", "-o", exe + ".tmp"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    os.remove(gen)
    if r.returncode != 0:
        raise RuntimeError("team emulator build failed:\n" + r.stdout[-6000:])
    os.replace(exe + ".tmp", exe)
    return exe, team_threads(src)


def run(header: bytes, inputs: Sequence[bytes], decode: bool = False, waves: int = 4, out_cap: int | None = None, dual: bool = False,
        team: bool = False):
    """Code every input as one block (one wavefront each; dual: the decoder with two blocks per wavefront; team: the
    lockstep decoder).  Returns [(bytes, status, consumed)]."""
    if team:
        exe, thr = build_team(header)
        waves, dual = thr // 64, False
    else:
        exe = build_dual(header) if dual else build(header, waves)
    cap = out_cap if out_cap is not None else max(len(x) for x in inputs) + 4096
    with tempfile.TemporaryDirectory(dir=BUILD) as td:
        hp = os.path.join(td, "h.bin")
        open(hp, "wb").write(header)
        paths = []
        for i, d in enumerate(inputs):
            p = os.path.join(td, f"in{i}")
            open(p, "wb").write(bytes(d))
            paths.append(p)
        r = subprocess.run([exe, "dec3" if team else ("dec2" if dual else ("dec" if decode else "enc")), str(waves), hp, str(cap), os.path.join(td, "out"), *paths],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        if r.returncode != 0:
            raise RuntimeError(f"emulator failed ({r.returncode}): {r.stderr[-2000:]}")
        res = []
        for i in range(len(inputs)):
            line = [l for l in r.stdout.splitlines() if l.startswith(f"block {i} ")][0].split()
            status, consumed = int(line[3]), int(line[7])
            res.append((open(os.path.join(td, f"out.{i}"), "rb").read(), status, consumed))
        return res


# ---- the pipelined encoder (zpaq_amd/csrc/device/pipe_kernel.h) ----
def pipe_source(header: bytes, chunk: int | None = None, group: int | None = None, mode: int = 0) -> str:
    """Generated source of the pipelined encoder: mode 0 = throughput (lane per block, SSE per bit position),
    1 = latency (MIX / CM / MIX2 per bit position as well); chunk / group: bytes per step / blocks per wavefront."""
    import zpaq_amd as z
    L = z.lib()
    L.zpq_plan_pipe_source_opts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p]
    plan = z.Plan(header)
    buf = C.create_string_buffer(4 << 20)
    ln = C.c_size_t(0)
    key = C.create_string_buffer(41)
    rc = L.zpq_plan_pipe_source_opts(plan._h, int(mode), int(chunk or 0), int(group or 0), buf, len(buf), C.byref(ln), key)
    if rc != 0:
        raise RuntimeError(L.zpq_last_error().decode())
    return buf.value.decode()


class _env:
    def __init__(self, **kv):
        self.kv = {k: (None if v is None else str(v)) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def pipe_build(header: bytes, chunk: int | None = None, group: int | None = None, mode: int = 0) -> str:
    import zpaq_amd as z
    src = pipe_source(header, chunk, group, mode)
    dev = os.path.join(ROOT, "zpaq_amd", "csrc", "device")
    deps = b"".join(open(p, "rb").read() for p in (
        os.path.join(EMU, "wave_emu.h"), os.path.join(EMU, "wave_emu.cpp"), os.path.join(EMU, "pipe_emu_main.cpp"), os.path.join(EMU, "guard_alloc.h"),
        os.path.join(dev, "pipe_kernel.h"), os.path.join(dev, "pipe_persist.h"), os.path.join(dev, "spec_kernel.h"), os.path.join(dev, "layout.h")))
    extra_flags = _sanitize_flags()
    key = hashlib.sha1(src.encode() + deps + " ".join(extra_flags).encode()).hexdigest()[:20]
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, f"pipe_{key}")
    if os.path.exists(exe):
        return exe
    gen = os.path.join(BUILD, f"pgen_{key}.cpp")
    with open(gen, "w") as fh:
        fh.write('#include "wave_emu.h"\n' + src)
    libdir = os.path.dirname(z.library_path())
    cmd = ["g++", "-O1", "-std=c++17", "-w", *extra_flags, "-I", EMU, "-I", dev, "-I", os.path.join(ROOT, "include"), gen,
           os.path.join(EMU, "pipe_emu_main.cpp"), os.path.join(EMU, "wave_emu.cpp"),
           "-L", libdir, "-lzpaq_amd", f"-Wl,-rpath,{libdir}", "-o", exe + ".tmp"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    os.remove(gen)
    if r.returncode != 0:
        raise RuntimeError("pipe emulator build failed:\n" + r.stdout[-6000:])
    os.replace(exe + ".tmp", exe)
    return exe


def pipe_run(header: bytes, inputs: Sequence[bytes], chunk: int | None = 64, out_cap: int | None = None,
             group: int | None = None, mode: int = 0, persist: bool = False):
    """Encode every input as one block with the pipelined encoder.  Returns [(bytes, status, consumed)].
    persist: the persistent launch (device/pipe_persist.h) -- every workgroup of the grid alive at once, the units waiting
    for each other through their progress counters -- instead of the six kernels step by step."""
    exe = pipe_build(header, chunk, group, mode)
    cap = out_cap if out_cap is not None else max(len(x) for x in inputs) + 4096
    with tempfile.TemporaryDirectory(dir=BUILD) as td:
        hp = os.path.join(td, "h.bin")
        open(hp, "wb").write(header)
        paths = []
        for i, d in enumerate(inputs):
            p = os.path.join(td, f"in{i}")
            open(p, "wb").write(bytes(d))
            paths.append(p)
        r = subprocess.run([exe, hp, str(cap), os.path.join(td, "out"), str(int(mode) | (16 if persist else 0)), str(int(chunk or 0)), str(int(group or 0)), *paths],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1800)
        if r.returncode != 0:
            raise RuntimeError(f"pipe emulator failed ({r.returncode}): {r.stderr[-2000:]}")
        res = []
        for i in range(len(inputs)):
            line = [l for l in r.stdout.splitlines() if l.startswith(f"block {i} ")][0].split()
            status, consumed = int(line[3]), int(line[7])
            res.append((open(os.path.join(td, f"out.{i}"), "rb").read(), status, consumed))
        return res


def build_lz77() -> str:
    """The emulator executable of the pre-processor kernels behind the suffix sort (device/lz77_kernel.h)."""
    import zpaq_amd as z
    dev = os.path.join(ROOT, "zpaq_amd", "csrc", "device")
    srcs = (os.path.join(EMU, "wave_emu.h"), os.path.join(EMU, "wave_emu.cpp"), os.path.join(EMU, "lz77_emu_main.cpp"),
            os.path.join(dev, "lz77_kernel.h"), os.path.join(dev, "layout.h"))
    flags = _sanitize_flags()
    key = hashlib.sha1(b"".join(open(p, "rb").read() for p in srcs) + " ".join(flags).encode()).hexdigest()[:20]
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, f"lz77_{key}")
    if os.path.exists(exe):
        return exe
    libdir = os.path.dirname(z.library_path())
    cmd = ["g++", "-O1", "-std=c++17", "-w", *flags, "-I", EMU, "-I", dev, "-I", os.path.join(ROOT, "include"),
           os.path.join(EMU, "lz77_emu_main.cpp"), os.path.join(EMU, "wave_emu.cpp"), "-L", libdir, "-lzpaq_amd", f"-Wl,-rpath,{libdir}", "-o", exe + ".tmp"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("lz77 emulator build failed:\n" + r.stdout[-6000:])
    os.replace(exe + ".tmp", exe)
    return exe


def lz77_run(kind: int, min_match: int, lookahead: int, bucket: int, checkbits: int, inputs: Sequence[bytes]):
    """device/lz77_kernel.h on the emulator: per input the token list (kind 1 / 2: 16 bytes per match) or the BWT stream
    (kind 3: n + 5 bytes) the device would hand back."""
    exe = build_lz77()
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for k, data in enumerate(inputs):
            pth = os.path.join(td, f"in{k}")
            with open(pth, "wb") as fh:
                fh.write(bytes(data))
            paths.append(pth)
        prefix = os.path.join(td, "out")
        r = subprocess.run([exe, str(kind), str(min_match), str(lookahead), str(bucket), str(checkbits), prefix, *paths],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"lz77 emulator failed ({r.returncode}):\n" + r.stdout[-4000:])
        return [open(f"{prefix}.{k}", "rb").read() for k in range(len(inputs))]
