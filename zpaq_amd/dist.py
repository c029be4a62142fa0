"""Multi-GPU block distribution: one process per GPU, blocks are independent.

ZPAQ blocks never exchange data while being coded, so the hot path has no collective.  The only
communication is moving blocks to the ranks that code them and moving the archives back:

  shard_range      block b of B goes to rank b*W//B (contiguous ranges keep archive order trivial)
  scatter_blocks   root holds [B, S] bytes -> every rank gets its [B_r, S] slice
  gather_archives  variable-length archives -> root, concatenated in block order
  max_over_ranks   the timing reduction bench.py needs

Backend: "nccl" (= RCCL over xGMI) with device tensors on the GPU box; "gloo" with CPU tensors in
the CPU tests (tests/test_dist.py, world_size 2).  With W == 1 everything degenerates to a no-op.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(nblocks: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of the blocks rank `rank` codes."""
    return nblocks * rank // world, nblocks * (rank + 1) // world


def _dev():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def scatter_blocks(blocks, nblocks: int, block_bytes: int, root: int = 0) -> torch.Tensor:
    """Root passes the whole corpus [nblocks, block_bytes] (uint8: a numpy array, or a torch tensor that may already be
    on the device); every rank receives its slice.

    Grouped point-to-point sends (ncclSend/ncclRecv under RCCL): per-rank slices differ in size
    when nblocks is not a multiple of the world size, which scatter() cannot express.
    """
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return blocks if isinstance(blocks, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(blocks))
    rank, world = dist.get_rank(), dist.get_world_size()
    b, e = shard_range(nblocks, rank, world)
    dev = _dev()
    mine = torch.empty((e - b, block_bytes), dtype=torch.uint8, device=dev)
    if rank == root:
        full = (blocks if isinstance(blocks, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(blocks))).to(dev)
        ops = []
        for r in range(world):
            rb, re = shard_range(nblocks, r, world)
            if r == root:
                mine.copy_(full[rb:re])
            elif re > rb:
                ops.append(dist.P2POp(dist.isend, full[rb:re].contiguous(), r))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
    elif e > b:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, mine, root)]):
            w.wait()
    return mine


def gather_archives(local: Sequence[bytes], root: int = 0) -> Optional[List[bytes]]:
    """Every rank passes the archives of its blocks (in block order); root gets all of them in order."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local)
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = _dev()
    # 1. how many archives / bytes each rank holds
    meta = torch.tensor([len(local), sum(len(a) for a in local)], dtype=torch.int64, device=dev)
    metas = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(metas, meta)
    counts = [int(m[0]) for m in metas]
    sizes = [int(m[1]) for m in metas]
    # 2. lengths and payloads, point to point to root
    lens = torch.tensor([len(a) for a in local], dtype=torch.int64, device=dev)
    payload = torch.from_numpy(np.frombuffer(b"".join(local), dtype=np.uint8).copy()).to(dev) if sizes[rank] else \
        torch.empty(0, dtype=torch.uint8, device=dev)
    if rank != root:
        ops = []
        if counts[rank]:
            ops.append(dist.P2POp(dist.isend, lens, root))
        if sizes[rank]:
            ops.append(dist.P2POp(dist.isend, payload, root))
        for w in (dist.batch_isend_irecv(ops) if ops else []):
            w.wait()
        return None
    all_lens = [lens if r == root else torch.empty(counts[r], dtype=torch.int64, device=dev) for r in range(world)]
    all_pay = [payload if r == root else torch.empty(sizes[r], dtype=torch.uint8, device=dev) for r in range(world)]
    ops = []
    for r in range(world):
        if r == root:
            continue
        if counts[r]:
            ops.append(dist.P2POp(dist.irecv, all_lens[r], r))
        if sizes[r]:
            ops.append(dist.P2POp(dist.irecv, all_pay[r], r))
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()
    out: List[bytes] = []
    for r in range(world):
        buf = all_pay[r].cpu().numpy().tobytes()
        pos = 0
        for n in all_lens[r].cpu().tolist():
            out.append(buf[pos:pos + n])
            pos += n
    return out


def max_over_ranks(seconds: float) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=_dev())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
