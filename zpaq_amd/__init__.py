"""zpaq_amd -- MI355X-native ZPAQ context-mixing coder (hot path of libzpaq 7.15).

Python is plumbing here: this module is a ctypes mirror of the C ABI in
``include/zpaq_amd.h`` (same entry points, same argument meaning, same error
behaviour), used by the tests and by ``bench.py``.  The product is
``zpaq_amd/libzpaq_amd.so`` (HIP kernels for gfx950 + C++ host library, built
in-tree by ``zpaq_amd/csrc/Makefile``).

There is no CPU fallback: if the shared library is missing, or no gfx950 device
is present, the modelled path raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

__all__ = [
    "ZpaqError", "lib", "library_path", "version", "init", "device_count", "shutdown", "set_kernel",
    "set_state_budget", "Plan", "encode_batch", "decode_batch", "compress_blocks", "compress_block",
    "decompress", "sha1", "expand_method", "method_to_header", "assemble", "table", "selftest",
    "last_timing", "STATUS",
]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBNAME = "libzpaq_amd.so"
_u8p = C.POINTER(C.c_ubyte)

STATUS = {0: "OK", 1: "NOMEM", 2: "CORRUPT", 3: "OVERFLOW", 4: "HEADER", 5: "VM", 6: "EOF", 7: "DEVICE",
          8: "UNSUPPORTED", 9: "ARG"}


class ZpaqError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[{STATUS.get(code, code)}] {msg}")
        self.code = code


def library_path() -> str:
    # ZPAQ_AMD_LIB selects an alternative in-tree build (e.g. the cycle-profiling variant)
    return os.path.join(_HERE, os.environ.get("ZPAQ_AMD_LIB", _LIBNAME))


_lib = None


def lib():
    """The loaded C-ABI library; raises loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `make -C zpaq_amd/csrc` (or __graft_entry__.build()). "
            "zpaq_amd has no pure-Python or CPU fallback for the coder.")
    L = C.CDLL(path)
    L.zpq_last_error.restype = C.c_char_p
    L.zpq_version.restype = C.c_char_p
    L.zpq_plan_create.argtypes = [_u8p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.zpq_plan_destroy.argtypes = [C.c_void_p]
    L.zpq_plan_ncomp.argtypes = [C.c_void_p]
    L.zpq_plan_memory.argtypes = [C.c_void_p]
    L.zpq_plan_memory.restype = C.c_double
    L.zpq_plan_state_bytes.argtypes = [C.c_void_p]
    L.zpq_plan_state_bytes.restype = C.c_uint64
    L.zpq_plan_algo_bytes_per_byte.argtypes = [C.c_void_p]
    L.zpq_plan_algo_bytes_per_byte.restype = C.c_double
    L.zpq_set_state_budget.argtypes = [C.c_uint64]
    L.zpq_sha1.argtypes = [_u8p, C.c_uint64, _u8p]
    L.zpq_sha1.restype = None
    L.zpq_table.restype = C.c_size_t
    L.zpq_table.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
    L.zpq_decompress.argtypes = [_u8p, C.c_uint64, _u8p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.zpq_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                    C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                    C.c_void_p, C.c_void_p, C.c_int]
    L.zpq_decode_device.argtypes = L.zpq_encode_device.argtypes
    _lib = L
    return L


def _check(rc: int):
    if rc != 0:
        raise ZpaqError(rc, lib().zpq_last_error().decode("latin1"))


def _arr(b) -> np.ndarray:
    if isinstance(b, np.ndarray):
        return np.ascontiguousarray(b, dtype=np.uint8).reshape(-1)
    b = bytes(b)
    return np.frombuffer(b, dtype=np.uint8).copy() if len(b) else np.zeros(0, np.uint8)


def _p(a: np.ndarray):
    return a.ctypes.data_as(_u8p)


def version() -> str:
    return lib().zpq_version().decode()


def device_count() -> int:
    return int(lib().zpq_device_count())


def init(device: Optional[int] = None):
    """zpq_init: bind to one GPU (default LOCAL_RANK, else 0)."""
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    _check(lib().zpq_init(int(device)))


def shutdown():
    lib().zpq_shutdown()


def set_kernel(which: int):
    """0 auto, 1 generic one-lane kernel, 2 generic wave-parallel kernel, 3 per-header specialised kernel."""
    _check(lib().zpq_set_kernel(int(which)))


def set_state_budget(nbytes: int):
    _check(lib().zpq_set_state_budget(int(nbytes)))


def selftest() -> List[int]:
    out = (C.c_int32 * 8)()
    _check(lib().zpq_selftest(out))
    return list(out)


def last_timing() -> Tuple[float, float, int]:
    a, b, n = C.c_float(0), C.c_float(0), C.c_uint32(0)
    lib().zpq_last_timing(C.byref(a), C.byref(b), C.byref(n))
    return a.value, b.value, n.value


class Plan:
    """zpq_plan: a parsed block header (ZPAQL::read + Predictor::init sizing)."""

    def __init__(self, header):
        h = _arr(header)
        self.header = h.tobytes()
        self._h = C.c_void_p()
        _check(lib().zpq_plan_create(_p(h), h.size, C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().zpq_plan_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def ncomp(self) -> int:
        return int(lib().zpq_plan_ncomp(self._h))

    @property
    def memory(self) -> float:
        return float(lib().zpq_plan_memory(self._h))

    @property
    def state_bytes(self) -> int:
        return int(lib().zpq_plan_state_bytes(self._h))

    @property
    def algo_bytes_per_byte(self) -> float:
        return float(lib().zpq_plan_algo_bytes_per_byte(self._h))


def encode_batch(plans: Sequence[Plan], inputs: Sequence, out_cap: Optional[Sequence[int]] = None,
                 check: bool = True):
    """zpq_encode_batch -> list of coded byte strings (and statuses if check=False)."""
    n = len(inputs)
    if isinstance(plans, Plan):
        plans = [plans] * n
    ins = [_arr(x) for x in inputs]
    caps = [int(c) for c in out_cap] if out_cap is not None else [a.size + a.size // 4 + 4096 for a in ins]
    outs = [np.empty(max(c, 1), np.uint8) for c in caps]
    PA = (C.c_void_p * n)(*[p._h for p in plans])
    IA = (_u8p * n)(*[_p(a) for a in ins])
    OA = (_u8p * n)(*[_p(a) for a in outs])
    IL = (C.c_uint32 * n)(*[a.size for a in ins])
    OC = (C.c_uint32 * n)(*caps)
    OL = (C.c_uint32 * n)()
    ST = (C.c_int32 * n)()
    rc = lib().zpq_encode_batch(PA, IA, IL, n, OA, OC, OL, ST)
    if check:
        _check(rc)
        return [outs[i][:OL[i]].tobytes() for i in range(n)]
    return [outs[i][:min(OL[i], caps[i])].tobytes() for i in range(n)], list(ST), list(OL)


def decode_batch(plans: Sequence[Plan], payloads: Sequence, max_out: Sequence[int], check: bool = True):
    """zpq_decode_batch -> list of (decoded bytes incl. PP header, consumed)."""
    n = len(payloads)
    if isinstance(plans, Plan):
        plans = [plans] * n
    ins = [_arr(x) for x in payloads]
    caps = [int(c) for c in max_out]
    outs = [np.empty(max(c, 1), np.uint8) for c in caps]
    PA = (C.c_void_p * n)(*[p._h for p in plans])
    IA = (_u8p * n)(*[_p(a) for a in ins])
    OA = (_u8p * n)(*[_p(a) for a in outs])
    IL = (C.c_uint32 * n)(*[a.size for a in ins])
    OC = (C.c_uint32 * n)(*caps)
    OL = (C.c_uint32 * n)()
    CO = (C.c_uint32 * n)()
    ST = (C.c_int32 * n)()
    rc = lib().zpq_decode_batch(PA, IA, IL, n, OA, OC, OL, CO, ST)
    if check:
        _check(rc)
        return [(outs[i][:OL[i]].tobytes(), int(CO[i])) for i in range(n)]
    return [(outs[i][:OL[i]].tobytes(), int(CO[i])) for i in range(n)], list(ST)


def compress_blocks(blocks: Sequence, method: str, filenames: Optional[Sequence[Optional[str]]] = None,
                    comments: Optional[Sequence[Optional[str]]] = None, dosha1: bool = True) -> List[bytes]:
    """Batched libzpaq::compressBlock: one ZPAQ block (archive bytes) per input buffer."""
    n = len(blocks)
    ins = [_arr(x).copy() for x in blocks]      # the call may modify inputs in place (E8E9)
    caps = [a.size + a.size // 4 + 8192 for a in ins]
    outs = [np.empty(c, np.uint8) for c in caps]

    def cstrs(v):
        if v is None:
            return None
        return (C.c_char_p * n)(*[None if s is None else (s if isinstance(s, bytes) else str(s).encode()) for s in v])

    IA = (_u8p * n)(*[_p(a) for a in ins])
    IL = (C.c_uint32 * n)(*[a.size for a in ins])
    OA = (_u8p * n)(*[_p(a) for a in outs])
    OC = (C.c_uint64 * n)(*caps)
    OL = (C.c_uint64 * n)()
    rc = lib().zpq_compress_blocks(method.encode(), IA, IL, n, cstrs(filenames), cstrs(comments), int(dosha1),
                                   OA, OC, OL)
    if rc == 3:   # OVERFLOW: sizes are in OL, retry once with exact capacities
        caps = [int(x) for x in OL]
        outs = [np.empty(max(c, 1), np.uint8) for c in caps]
        OA = (_u8p * n)(*[_p(a) for a in outs])
        OC = (C.c_uint64 * n)(*caps)
        ins = [_arr(x).copy() for x in blocks]
        IA = (_u8p * n)(*[_p(a) for a in ins])
        rc = lib().zpq_compress_blocks(method.encode(), IA, IL, n, cstrs(filenames), cstrs(comments), int(dosha1),
                                       OA, OC, OL)
    _check(rc)
    return [outs[i][:OL[i]].tobytes() for i in range(n)]


def compress_block(data, method: str, filename: Optional[str] = None, comment: Optional[str] = None,
                   dosha1: bool = True) -> bytes:
    return compress_blocks([data], method, [filename], [comment], dosha1)[0]


def decompress(archive, cap: Optional[int] = None) -> bytes:
    """libzpaq::decompress over a whole archive (all blocks decoded as one device batch)."""
    a = _arr(archive)
    cap = int(cap) if cap is not None else max(4 * a.size, 1 << 20)
    while True:
        out = np.empty(max(cap, 1), np.uint8)
        n = C.c_uint64(0)
        rc = lib().zpq_decompress(_p(a), a.size, _p(out), out.size, C.byref(n))
        if rc == 3 and n.value > out.size:
            cap = int(n.value)
            continue
        _check(rc)
        return out[:n.value].tobytes()


# ---- host-side pieces -------------------------------------------------------
def sha1(data) -> bytes:
    a = _arr(data)
    out = np.empty(20, np.uint8)
    lib().zpq_sha1(_p(a), a.size, _p(out))
    return out.tobytes()


def expand_method(method: str, data) -> str:
    a = _arr(data)
    buf = C.create_string_buffer(4096)
    _check(lib().zpq_expand_method(method.encode(), _p(a), C.c_uint32(a.size), buf, C.c_size_t(len(buf))))
    return buf.value.decode()


def _two_bufs(fn, first, args):
    h = np.empty(1 << 17, np.uint8)
    p = np.empty(1 << 17, np.uint8)
    hl, pl = C.c_size_t(0), C.c_size_t(0)
    _check(fn(first, args, _p(h), C.c_size_t(h.size), C.byref(hl), _p(p), C.c_size_t(p.size), C.byref(pl)))
    return h[:hl.value].tobytes(), p[:pl.value].tobytes()


def method_to_header(xmethod: str):
    """makeConfig + Compiler: "x.." -> (stored header bytes, pcomp bytes, args[9])."""
    args = (C.c_int * 9)()
    h, p = _two_bufs(lib().zpq_method_to_header, xmethod.encode(), args)
    return h, p, list(args)


def builtin_model_header(level: int) -> bytes:
    """Compressor::startBlock(int level): the stored header of min.cfg / mid.cfg / max.cfg (level 1 / 2 / 3)."""
    h = np.empty(4096, np.uint8)
    hl = C.c_size_t(0)
    _check(lib().zpq_builtin_model_header(C.c_int(level), _p(h), C.c_size_t(h.size), C.byref(hl)))
    return h[:hl.value].tobytes()


def assemble(config: str, args: Optional[Iterable[int]] = None):
    """Compiler: ZPAQL source -> (stored header bytes, pcomp bytes)."""
    a9 = (C.c_int * 9)(*(list(args or []) + [0] * 9)[:9])
    return _two_bufs(lib().zpq_assemble, config.encode(), a9)


_TABLES = {"squash": (0, np.uint16, 4096), "stretch": (1, np.int16, 32768), "dt": (2, np.int32, 1024),
           "dt2k": (3, np.int32, 256), "state": (4, np.uint8, 1024)}


def table(name: str) -> np.ndarray:
    which, dt, n = _TABLES[name]
    out = np.empty(n, dt)
    got = lib().zpq_table(which, out.ctypes.data_as(C.c_void_p), out.nbytes)
    if got != out.nbytes:
        raise ZpaqError(7, f"table {name} unavailable")
    return out
