"""Deterministic, integer-only synthetic corpora.

Every generator is a pure function of (kind, nbytes, seed) built from the 32-bit
LCG ``x = x*1664525 + 1013904223`` so that blocks are bit-identical on every
platform; block ``b`` of a corpus uses ``seed = 12345 + b``.  The LCG stream
itself is the one BASELINE.md §2 quotes known-answer archive hashes for
(byte = ``x >> 24``, first byte after one step from ``x0``).

Kinds
  zeros    all-zero block                         (highly compressible)
  pattern  one repeated 64-byte LCG pattern       (highly compressible)
  lcg      LCG bytes                              (incompressible / "random")
  text     "enwik-style" Zipf text, 4096-word vocabulary, rank ~ 1/r
  records  16-byte binary records (u32 counter, u32 LCG, u64 slow walk) --
           triggers level-5 period detection (libzpaq.cpp:7655-7688)
  mixed    by seed: text, text, lcg, records      (config C4)
"""
from __future__ import annotations

import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)
BASE_SEED = 12345


def lcg_u32(n: int, seed: int) -> np.ndarray:
    """x_1..x_n of the LCG started at x_0 = seed (uint32 array), vectorised.

    x_k = a^k x_0 + c (a^{k-1} + ... + 1)  (mod 2^32); both factors are prefix
    products/sums that wrap mod 2^32 in uint32 arithmetic.
    """
    if n <= 0:
        return np.zeros(0, dtype=np.uint32)
    with np.errstate(over="ignore"):
        apow = np.cumprod(np.full(n, _A, dtype=np.uint32), dtype=np.uint32)  # a^1..a^n
        geo = np.empty(n, dtype=np.uint32)  # 1 + a + ... + a^{k-1}
        geo[0] = 1
        if n > 1:
            geo[1:] = np.cumsum(apow[:-1], dtype=np.uint32) + np.uint32(1)
        return apow * np.uint32(seed & 0xFFFFFFFF) + _C * geo


def lcg_bytes(n: int, seed: int) -> np.ndarray:
    return (lcg_u32(n, seed) >> np.uint32(24)).astype(np.uint8)


_VOCAB = None


def _vocab():
    """4096 lowercase words of length 2..10, drawn once from a fixed LCG stream."""
    global _VOCAB
    if _VOCAB is None:
        r = lcg_u32(4096 * 11, 0x5EED) >> np.uint32(16)
        r = r.reshape(4096, 11)
        lens = (2 + (r[:, 0] % 9)).astype(np.int64)
        letters = (97 + (r[:, 1:] % 26)).astype(np.uint8)  # [4096,10]
        # integer Zipf CDF: weight(rank r) = floor(2^32/(r+1))
        w = (np.uint64(1) << np.uint64(32)) // np.arange(1, 4097, dtype=np.uint64)
        cdf = np.cumsum(w, dtype=np.uint64)
        _VOCAB = (lens, letters, cdf)
    return _VOCAB


def zipf_text(n: int, seed: int) -> np.ndarray:
    lens, letters, cdf = _vocab()
    total = int(cdf[-1])
    # mean word length under this CDF is < 7 incl. separator >= 3: draw n/3+16 words
    nw = n // 3 + 16
    r = lcg_u32(2 * nw, seed)
    u = (r[0::2] >> np.uint32(4)).astype(np.uint64)  # 28 random bits
    target = (u * np.uint64(total)) >> np.uint64(28)
    idx = np.searchsorted(cdf, target, side="right").astype(np.int64)
    idx = np.minimum(idx, 4095)
    sep = np.where((r[1::2] >> np.uint32(28)) == 0, 10, 32).astype(np.uint8)
    wl = lens[idx]
    tot = wl + 1
    ends = np.cumsum(tot)
    starts = ends - tot
    size = int(ends[-1])
    out = np.empty(size, dtype=np.uint8)
    # position within word for every output byte
    word_of = np.repeat(np.arange(nw, dtype=np.int64), tot)
    pos = np.arange(size, dtype=np.int64) - starts[word_of]
    is_sep = pos == wl[word_of]
    out[:] = letters[idx[word_of], np.minimum(pos, 9)]
    out[is_sep] = sep[word_of[is_sep]]
    assert size >= n
    return out[:n].copy()


def records(n: int, seed: int) -> np.ndarray:
    nr = (n + 15) // 16
    r = lcg_u32(2 * nr, seed)
    rec = np.zeros((nr, 16), dtype=np.uint8)
    ctr = (np.arange(nr, dtype=np.uint64) + np.uint64(seed & 0xFFFF)).astype(np.uint32)
    rec[:, 0:4] = ctr.view(np.uint8).reshape(nr, 4)
    rec[:, 4:8] = r[0::2].copy().view(np.uint8).reshape(nr, 4)
    walk = np.cumsum((r[1::2] >> np.uint32(29)).astype(np.uint64), dtype=np.uint64)
    walk = walk + np.uint64(0x0102030405060000)
    rec[:, 8:16] = walk.view(np.uint8).reshape(nr, 8)
    return rec.reshape(-1)[:n].copy()


def block(kind: str, n: int, seed: int) -> np.ndarray:
    """One block of `n` bytes (uint8 array)."""
    if kind == "zeros":
        return np.zeros(n, dtype=np.uint8)
    if kind == "pattern":
        pat = lcg_bytes(64, seed)
        return np.tile(pat, (n + 63) // 64)[:n].copy()
    if kind == "lcg":
        return lcg_bytes(n, seed)
    if kind == "text":
        return zipf_text(n, seed)
    if kind == "records":
        return records(n, seed)
    if kind == "mixed":
        return block(("text", "text", "lcg", "records")[(seed - BASE_SEED) % 4], n, seed)
    raise ValueError(f"unknown corpus kind {kind!r}")


def corpus(kind: str, nblocks: int, block_bytes: int, first_block: int = 0) -> np.ndarray:
    """[nblocks, block_bytes] uint8; block b uses seed 12345 + first_block + b."""
    out = np.empty((nblocks, block_bytes), dtype=np.uint8)
    for b in range(nblocks):
        out[b] = block(kind, block_bytes, BASE_SEED + first_block + b)
    return out
