"""ctypes bindings for the CHECKERS under oracle/ -- test infrastructure only.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (zpaq_amd/) never does.

  Ref     -> oracle/_ref/libzpaq_ref.so : the unmodified reference library built
             from /root/reference by oracle/Makefile (kind "reference").
  Oracle  -> oracle/libzpaq_oracle.so   : our plain-C restatement of the hot path
             (kind "port").
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_ubyte)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


def _bytes_arr(b) -> np.ndarray:
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8)
    return np.frombuffer(bytes(b), dtype=np.uint8).copy() if len(b) else np.zeros(0, np.uint8)


def _cpu_has_v3() -> bool:
    """AVX2 + BMI2 + FMA on the CPU this runs on (the -march=x86-64-v3 build of the reference needs them)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("flags"):
                f = set(ln.split(":", 1)[1].split())
                return {"avx2", "bmi2", "fma", "movbe", "f16c"} <= f and ("abm" in f or "lzcnt" in f)
    except Exception:
        pass
    return False


class Ref:
    """The compiled reference (JIT build by default, nojit=True for the interpreter)."""

    def divsufsort(self, data) -> "np.ndarray":
        """The suffix array libzpaq's LZBuffer sorts with (divsufsort, libzpaq.cpp:4658-6434), as int32[n]."""
        import numpy as np
        if self._divsufsort is None:
            raise RuntimeError("the compiled reference does not export divsufsort")
        a = np.frombuffer(bytes(data), np.uint8).copy() if len(data) else np.zeros(1, np.uint8)
        sa = np.zeros(max(len(data), 1), np.int32)
        rc = self._divsufsort(a.ctypes.data_as(_u8p), sa.ctypes.data_as(C.POINTER(C.c_int)), len(data))
        if rc != 0:
            raise RuntimeError("divsufsort failed: %d" % rc)
        return sa[:len(data)]

    def __init__(self, nojit: bool = False):
        name = "libzpaq_ref_nojit.so" if nojit else "libzpaq_ref.so"
        self.flags = "-O3 -Dunix" + (" -DNOJIT" if nojit else "")
        if not nojit and os.path.exists(os.path.join(_HERE, "_ref", "libzpaq_ref_v3.so")) and _cpu_has_v3():
            name = "libzpaq_ref_v3.so"
            self.flags = "-O3 -march=x86-64-v3 -Dunix"
        path = os.path.join(_HERE, "_ref", name)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` where /root/reference exists")
        L = self.lib = C.CDLL(path)
        L.ref_last_error.restype = C.c_char_p
        L.ref_compress_block.restype = C.c_longlong
        L.ref_compress_block.argtypes = [_u8p, C.c_size_t, C.c_char_p, C.c_char_p, C.c_char_p,
                                         C.c_int, _u8p, C.c_size_t]
        L.ref_compress.restype = C.c_longlong
        L.ref_compress.argtypes = L.ref_compress_block.argtypes
        L.ref_decompress.restype = C.c_longlong
        L.ref_decompress.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t]
        L.ref_compress_level.restype = C.c_longlong
        L.ref_compress_level.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_char_p, C.c_char_p,
                                         C.c_int, _u8p, C.c_size_t]
        L.ref_compress_config.restype = C.c_longlong
        L.ref_compress_config.argtypes = [_u8p, C.c_size_t, C.c_char_p, C.POINTER(C.c_int),
                                          C.c_char_p, C.c_char_p, C.c_int, _u8p, C.c_size_t]
        # the reference's suffix sorter itself (libzpaq.cpp:6371, an external symbol of the compiled reference)
        self._divsufsort = getattr(L, "_ZN7libzpaq10divsufsortEPKhPii", None)
        if self._divsufsort is not None:
            self._divsufsort.restype = C.c_int
            self._divsufsort.argtypes = [_u8p, C.POINTER(C.c_int), C.c_int]
        L.ref_make_config.restype = C.c_longlong
        L.ref_make_config.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        L.ref_compile.restype = C.c_longlong
        L.ref_compile.argtypes = [C.c_char_p, C.POINTER(C.c_int), _u8p, C.c_size_t, _u8p,
                                  C.c_size_t, C.POINTER(C.c_longlong)]
        L.ref_sha1.restype = None
        L.ref_sha1.argtypes = [_u8p, C.c_size_t, _u8p]
        L.ref_state_table.restype = None
        L.ref_state_table.argtypes = [_u8p]
        L.ref_block_memory.restype = C.c_double
        L.ref_block_memory.argtypes = [_u8p, C.c_size_t]
        L.ref_compress_blocks_mt.restype = C.c_double
        L.ref_compress_blocks_mt.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_char_p, C.c_int,
                                             C.POINTER(C.c_longlong), _u8p, C.c_size_t, C.c_double]

    def _err(self):
        return RuntimeError(self.lib.ref_last_error().decode("latin1"))

    def build_flags(self) -> str:
        """Compiler flags of the reference build that was loaded (for the cpu_baseline line)."""
        return self.flags

    @staticmethod
    def _s(x):
        return None if x is None else (x if isinstance(x, bytes) else str(x).encode())

    def _call_out(self, fn, data, cap, *mid):
        a = _bytes_arr(data)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        r = fn(_ptr(a), a.size, *mid, _ptr(out), out.size)
        if r < 0:
            raise self._err()
        if r > out.size:
            return self._call_out(fn, data, int(r), *mid)
        return out[:r].tobytes()

    def compress_block(self, data, method, filename=None, comment=None, dosha1=True) -> bytes:
        n = len(data)
        return self._call_out(self.lib.ref_compress_block, data, n + n // 4 + 4096,
                              self._s(method), self._s(filename), self._s(comment), int(dosha1))

    def compress(self, data, method, filename=None, comment=None, dosha1=True) -> bytes:
        n = len(data)
        return self._call_out(self.lib.ref_compress, data, n + n // 4 + 65536,
                              self._s(method), self._s(filename), self._s(comment), int(dosha1))

    def compress_level(self, data, level, filename=None, comment=None, dosha1=True) -> bytes:
        n = len(data)
        return self._call_out(self.lib.ref_compress_level, data, n + n // 4 + 4096,
                              int(level), self._s(filename), self._s(comment), int(dosha1))

    def compress_level_segments(self, parts, level) -> bytes:
        """one block, built-in model `level`, one segment per part (named s0, s1, ..; SHA-1 trailers)"""
        parts = [_bytes_arr(p) for p in parts]
        allb = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
        lens = (C.c_size_t * len(parts))(*[p.size for p in parts])
        cap = allb.size * 2 + 65536
        out = np.empty(cap, np.uint8)
        self.lib.ref_compress_level_segments.restype = C.c_longlong
        self.lib.ref_compress_level_segments.argtypes = [_u8p, C.POINTER(C.c_size_t), C.c_int, C.c_int, _u8p, C.c_size_t]
        r = self.lib.ref_compress_level_segments(_ptr(allb), lens, len(parts), int(level), _ptr(out), cap)
        if r < 0:
            raise self._err()
        return out[:r].tobytes()

    def compress_config(self, data, config, args=None, filename=None, comment=None, dosha1=True) -> bytes:
        n = len(data)
        a9 = (C.c_int * 9)(*(list(args or []) + [0] * 9)[:9])
        return self._call_out(self.lib.ref_compress_config, data, n + n // 4 + 4096,
                              self._s(config), a9, self._s(filename), self._s(comment), int(dosha1))

    def decompress(self, archive, cap) -> bytes:
        a = _bytes_arr(archive)
        out = np.empty(max(cap, 1), dtype=np.uint8)
        r = self.lib.ref_decompress(_ptr(a), a.size, _ptr(out), out.size)
        if r < 0:
            raise self._err()
        if r > out.size:
            return self.decompress(archive, int(r))
        return out[:r].tobytes()

    def make_config(self, method):
        args = (C.c_int * 9)()
        buf = C.create_string_buffer(1 << 20)
        r = self.lib.ref_make_config(self._s(method), args, buf, len(buf))
        if r < 0:
            raise self._err()
        return buf.raw[:r].decode("latin1"), list(args)

    def compile(self, config, args=None):
        """-> (hcomp header bytes as stored in the archive, pcomp bytes incl. len16 or b'')."""
        a9 = (C.c_int * 9)(*(list(args or []) + [0] * 9)[:9])
        h = np.empty(1 << 17, np.uint8)
        p = np.empty(1 << 17, np.uint8)
        pl = C.c_longlong(0)
        r = self.lib.ref_compile(self._s(config), a9, _ptr(h), h.size, _ptr(p), p.size, C.byref(pl))
        if r < 0:
            raise self._err()
        return h[:r].tobytes(), p[:pl.value].tobytes()

    def sha1(self, data) -> bytes:
        a = _bytes_arr(data)
        out = np.empty(20, np.uint8)
        self.lib.ref_sha1(_ptr(a), a.size, _ptr(out))
        return out.tobytes()

    def state_table(self) -> np.ndarray:
        out = np.empty(1024, np.uint8)
        self.lib.ref_state_table(_ptr(out))
        return out

    def block_memory(self, archive) -> float:
        a = _bytes_arr(archive)
        return float(self.lib.ref_block_memory(_ptr(a), a.size))

    def compress_blocks_mt(self, blocks: np.ndarray, method, nthreads: int, deadline_s: float = 0.0, keep: bool = False):
        """blocks [nblocks, block_bytes] uint8 -> (wall seconds, archive sizes; -2 = not started
        because deadline_s had passed); keep=True also returns the archives (list of bytes / None)."""
        blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
        nb, bs = blocks.shape
        lens = (C.c_longlong * nb)()
        out = stride = None
        if keep:
            stride = bs + bs // 2 + 4096
            out = np.empty((nb, stride), np.uint8)
        s = self.lib.ref_compress_blocks_mt(_ptr(blocks), bs, nb, self._s(method), int(nthreads),
                                            lens, _ptr(out) if keep else None, stride or 0, float(deadline_s))
        if s < 0:
            raise self._err()
        if keep:
            return s, list(lens), [out[b, :lens[b]].tobytes() if 0 <= lens[b] <= stride else None for b in range(nb)]
        return s, list(lens)


    def decompress_blocks_mt(self, archives, block_bytes: int, nthreads: int) -> float:
        """libzpaq::decompress over every archive (one ZPAQ block each) from an nthreads work queue; wall seconds."""
        n = len(archives)
        bufs = [np.frombuffer(a, dtype=np.uint8) for a in archives]
        PA = (C.c_void_p * n)(*[b.ctypes.data for b in bufs])
        LN = (C.c_size_t * n)(*[b.size for b in bufs])
        f = self.lib.ref_decompress_blocks_mt
        f.restype = C.c_double
        f.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.c_size_t, C.c_int]
        s = f(PA, LN, n, int(block_bytes), int(nthreads))
        if s < 0:
            raise self._err()
        return s


def have_ref() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libzpaq_ref.so"))


class Oracle:
    """Our plain-C restatement (oracle/zpaq_oracle.c)."""

    def __init__(self):
        path = os.path.join(_HERE, "libzpaq_oracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle libzpaq_oracle.so`")
        L = self.lib = C.CDLL(path)
        L.zo_tables_ok.restype = C.c_int
        for name, ty, n in (("zo_squash_table", C.c_uint16, 4096), ("zo_stretch_table", C.c_int16, 32768),
                            ("zo_dt_table", C.c_int32, 1024), ("zo_dt2k_table", C.c_int32, 256),
                            ("zo_state_table", C.c_uint8, 1024)):
            getattr(L, name).restype = C.POINTER(ty * n)
        L.zo_encode.restype = C.c_longlong
        L.zo_encode.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, _u8p, C.c_size_t,
                                C.POINTER(C.c_uint16), C.c_size_t]
        L.zo_decode.restype = C.c_longlong
        L.zo_decode.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, _u8p, C.c_size_t,
                                C.POINTER(C.c_size_t)]
        L.zo_hcomp_trace.restype = C.c_int
        L.zo_hcomp_trace.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.zo_model_new.restype = C.c_void_p
        L.zo_model_new.argtypes = [_u8p, C.c_size_t, C.POINTER(C.c_int)]
        L.zo_model_free.argtypes = [C.c_void_p]
        L.zo_model_memory.restype = C.c_double
        L.zo_model_memory.argtypes = [C.c_void_p]

    def tables_ok(self) -> bool:
        return bool(self.lib.zo_tables_ok())

    def table(self, name) -> np.ndarray:
        return np.array(getattr(self.lib, f"zo_{name}_table")().contents)

    def memory(self, header) -> float:
        h = _bytes_arr(header)
        err = C.c_int(0)
        m = self.lib.zo_model_new(_ptr(h), h.size, C.byref(err))
        if not m:
            raise RuntimeError(f"zo_model_new failed: {err.value}")
        v = self.lib.zo_model_memory(m)
        self.lib.zo_model_free(m)
        return float(v)

    def encode(self, header, data, ntrace=0):
        """-> coded bytes (incl. EOS flush), and optionally the first ntrace predictions."""
        h = _bytes_arr(header)
        d = _bytes_arr(data)
        cap = d.size + d.size // 2 + 4096
        out = np.empty(cap, np.uint8)
        tr = np.zeros(max(ntrace, 1), np.uint16)
        r = self.lib.zo_encode(_ptr(h), h.size, _ptr(d), d.size, _ptr(out), cap,
                               tr.ctypes.data_as(C.POINTER(C.c_uint16)), ntrace)
        if r < 0:
            raise RuntimeError(f"zo_encode failed: {r}")
        if r > cap:
            raise RuntimeError("zo_encode overflow")
        coded = out[:r].tobytes()
        return (coded, tr[:ntrace]) if ntrace else coded

    def decode(self, header, coded, cap):
        """-> (decoded bytes incl. PP header bytes, bytes of `coded` consumed)."""
        h = _bytes_arr(header)
        c = _bytes_arr(coded)
        out = np.empty(max(cap, 1), np.uint8)
        used = C.c_size_t(0)
        r = self.lib.zo_decode(_ptr(h), h.size, _ptr(c), c.size, _ptr(out), out.size, C.byref(used))
        if r < 0:
            raise RuntimeError(f"zo_decode failed: {r}")
        if r > out.size:
            return self.decode(header, coded, int(r))
        return out[:r].tobytes(), used.value

    def decode_outcome(self, header, coded, cap):
        """Decoder::decompress on a possibly damaged stream -> (status, bytes): status 0 with the decoded bytes, or
        2 ("archive corrupted", libzpaq.cpp:2108 / 2134) / 6 ("unexpected end of file", 2120) with b""."""
        h = _bytes_arr(header)
        c = _bytes_arr(coded)
        out = np.empty(max(cap, 1), np.uint8)
        used = C.c_size_t(0)
        r = self.lib.zo_decode(_ptr(h), h.size, _ptr(c), c.size, _ptr(out), out.size, C.byref(used))
        if r == -3:
            return 2, b""
        if r == -6:
            return 6, b""
        if r < 0:
            raise RuntimeError(f"zo_decode failed: {r}")
        return 0, out[:min(r, out.size)].tobytes()

    def hcomp_trace(self, header, data, ncomp) -> np.ndarray:
        h = _bytes_arr(header)
        d = _bytes_arr(data)
        out = np.zeros((d.size, ncomp), np.uint32)
        r = self.lib.zo_hcomp_trace(_ptr(h), h.size, _ptr(d), d.size,
                                    out.ctypes.data_as(C.POINTER(C.c_uint32)))
        if r < 0:
            raise RuntimeError(f"zo_hcomp_trace failed: {r}")
        return out


# ---- ZPAQ container parsing used by the tests to slice archives ----
TAG = bytes([0x37, 0x6B, 0x53, 0x74, 0xA0, 0x31, 0x83, 0xD3, 0x8C, 0xB2, 0x28, 0xB0, 0xD3])


def parse_block(archive: bytes, pos: int = 0) -> dict:
    """Split ONE single-segment block starting at `pos` (tag optional) into fields.

    Returns dict(header, filename, comment, payload_start, level, ...).  The
    coded payload's end is data dependent; use Oracle.decode()'s `consumed`.
    """
    a = archive
    if a[pos:pos + 13] == TAG:
        pos += 13
    assert a[pos:pos + 3] == b"zPQ", "no block"
    level, ztype = a[pos + 3], a[pos + 4]
    pos += 5
    hsize = a[pos] + 256 * a[pos + 1]
    header = a[pos:pos + hsize + 2]
    pos += hsize + 2
    assert a[pos] == 1, "no segment"
    pos += 1
    e = a.index(b"\0", pos)
    filename = a[pos:e]
    pos = e + 1
    e = a.index(b"\0", pos)
    comment = a[pos:e]
    pos = e + 1
    assert a[pos] == 0
    pos += 1
    return dict(level=level, type=ztype, header=header, filename=filename, comment=comment,
                payload_start=pos)
