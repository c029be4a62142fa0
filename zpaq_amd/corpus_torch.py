"""The "text" corpus of zpaq_amd.corpus generated on the GPU with torch (plumbing for bench.py:
1024 x 1 MiB Zipf-text blocks take minutes in numpy on a few host cores and under a second here).
Bit-identical to corpus.zipf_text -- same LCG stream, same integer CDF -- which
tests/test_gpu_parity.py::test_torch_corpus_matches_numpy checks."""
from __future__ import annotations

import numpy as np
import torch

from . import corpus

_A, _C = 1664525, 1013904223
_M32 = 0xFFFFFFFF


def _lcg_u32(n: int, seeds: torch.Tensor) -> torch.Tensor:
    """x_1..x_n for every seed: [len(seeds), n] int64 holding uint32 values.

    Prefix products/sums in int64 wrap mod 2^64, which preserves the low 32 bits."""
    dev = seeds.device
    a = torch.full((n,), _A, dtype=torch.int64, device=dev)
    apow = torch.cumprod(a, 0) & _M32                        # a^1..a^n mod 2^32
    geo = torch.ones(n, dtype=torch.int64, device=dev)
    if n > 1:
        geo[1:] = (torch.cumsum(apow[:-1], 0) + 1) & _M32     # 1 + a + .. + a^(k-1)
    x = (apow[None, :] * (seeds[:, None] & _M32)) & _M32
    return (x + ((_C * geo) & _M32)[None, :]) & _M32


def text_blocks(nblocks: int, n: int, first_seed: int, device, chunk: int = 32, seeds=None) -> torch.Tensor:
    """[nblocks, n] uint8 on `device`; block b == corpus.zipf_text(n, first_seed + b) -- or, with `seeds` (a list of
    nblocks ints), corpus.zipf_text(n, seeds[b]): the text blocks of the "mixed" corpus are not consecutive."""
    lens_np, letters_np, cdf_np = corpus._vocab()
    lens = torch.from_numpy(lens_np).to(device)
    letters = torch.from_numpy(letters_np.astype(np.int64)).to(device)            # [4096, 10]
    cdf = torch.from_numpy(cdf_np.astype(np.int64)).to(device)                    # < 2^45, fits int64
    total = int(cdf_np[-1])
    nw = n // 3 + 16
    out = torch.empty((nblocks, n), dtype=torch.uint8, device=device)
    for b0 in range(0, nblocks, chunk):
        k = min(chunk, nblocks - b0)
        if seeds is None:
            sd = torch.arange(first_seed + b0, first_seed + b0 + k, dtype=torch.int64, device=device)
        else:
            sd = torch.tensor(list(seeds[b0:b0 + k]), dtype=torch.int64, device=device)
        r = _lcg_u32(2 * nw, sd)                                                # [k, 2nw]
        u = r[:, 0::2] >> 4                                                        # 28 random bits
        # (u * total) >> 28 without leaving int64: total < 2^36, u < 2^28 (numpy does this in uint64)
        t1, t0 = total >> 14, total & 0x3FFF
        target = (u * t1 + ((u * t0) >> 14)) >> 14
        idx = torch.searchsorted(cdf, target.contiguous(), right=True).clamp_(max=4095)
        sep = torch.where((r[:, 1::2] >> 28) == 0, 10, 32).to(torch.uint8)
        wl = lens[idx]                                                             # [k, nw]
        tot = wl + 1
        ends = torch.cumsum(tot, 1)
        starts = ends - tot
        for j in range(k):                                                         # ragged: one block at a time
            size = int(ends[j, -1])
            word_of = torch.repeat_interleave(torch.arange(nw, device=device), tot[j], output_size=size)
            pos = torch.arange(size, device=device) - starts[j][word_of]
            wlj = wl[j][word_of]
            ch = letters[idx[j][word_of], pos.clamp(max=9)].to(torch.uint8)
            ch = torch.where(pos == wlj, sep[j][word_of], ch)
            out[b0 + j] = ch[:n]
        del r, u, target, idx, sep, wl, tot, ends, starts
    return out
