#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ZPAQ context-mixing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--blocks B --block-bytes S --kind text --method 5]

A "step" = one pass of the hot path (Predictor init + predict/update/encode of
every bit) over one batch of B synthetic blocks per GPU, inputs already resident
in HBM.  Default workload = BASELINE.json configs[2]: method "5" over 1024 x
1 MiB "enwik-style" Zipf text blocks on one MI355X.  N > 1 (launched by
torch.distributed.run, one rank per GPU): blocks are independent, so each rank
codes its own blocks with no data-path collective; ranks only barrier and
max-reduce the time.  --scaling weak (default): B blocks PER GPU (BASELINE
configs[3] is 8 x 1024); --scaling strong: B blocks in total, split over the
ranks.  Strong scaling of the 1024-block headline is limited by construction:
the per-bit chain of a block is serial and one GPU already runs all 1024 blocks
concurrently, so fewer blocks per GPU shorten the launch only as far as the
units of the pipelined encoder stop contending for the memory pipeline
(DESIGN.md section 6).

Prints ONE JSON line (rank 0).  `roofline` prices the coding launch sequence
against HBM (achieved = algorithmic model-state bytes per step / its duration
measured with hipEvents on the launch stream); `cpu_baseline` times the
reference libzpaq (oracle/_ref, kind "reference") or, if that was not built,
our C oracle (kind "port") on a bounded sample of the same blocks on this
box's host cores, and every archive the reference produced is compared with
ours by SHA-1; `api` is the same batch through the drop-in API on HOST buffers
(method expansion + SHA-1 + H2D + kernels + D2H + framing), outside the timed
region.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import struct
import json
import os
import re
import sys
import time

import numpy as np

# the encoder's seven streams want a hardware queue each (zpaq_amd/csrc/device/engine.cpp); read when HIP initialises,
# which torch does before the library is loaded
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def _gen_block(args):
    from zpaq_amd import corpus
    kind, nbytes, seed = args
    return corpus.block(kind, nbytes, seed)


def make_corpus(kind, nblocks, block_bytes, first, dev=None):
    """[nblocks, block_bytes] uint8, block b = corpus.block(kind, block_bytes, 12345 + first + b);
    kind "mixed" = BASELINE configs[3]: text, text, LCG-random, records by (first + b) mod 4.  With a torch device the
    text blocks (minutes of numpy on a few host cores) are generated there, bit-identical (corpus_torch)."""
    from zpaq_amd import corpus
    mix = ["text", "text", "lcg", "records"]
    kinds = [mix[(first + b) % 4] if kind == "mixed" else kind for b in range(nblocks)]
    out = np.empty((nblocks, block_bytes), np.uint8)
    on_dev = [b for b in range(nblocks) if kinds[b] == "text"] if dev is not None else []
    if on_dev:
        from zpaq_amd import corpus_torch
        for i0 in range(0, len(on_dev), 256):
            idx = on_dev[i0:i0 + 256]
            t = corpus_torch.text_blocks(len(idx), block_bytes, 0, dev, seeds=[corpus.BASE_SEED + first + b for b in idx])
            out[idx] = t.cpu().numpy()
            del t
    jobs = [(kinds[b], block_bytes, corpus.BASE_SEED + first + b) for b in range(nblocks) if b not in set(on_dev)]
    rest = [b for b in range(nblocks) if b not in set(on_dev)]
    nproc = min(max(len(jobs), 1), usable_cores(), 64)
    if nproc > 1 and len(jobs) * block_bytes >= (8 << 20):
        import multiprocessing as mp
        with mp.get_context("fork").Pool(nproc) as pool:
            for b, blk in zip(rest, pool.imap(_gen_block, jobs, chunksize=max(1, len(jobs) // (nproc * 4)))):
                out[b] = blk
    else:
        for b, j in zip(rest, jobs):
            out[b] = _gen_block(j)
    return out


def _size_name(n):
    for sh, u in ((20, "MiB"), (10, "KiB")):
        if n >= 1 << sh and n % (1 << sh) == 0:
            return f"{n >> sh} {u}"
    return f"{n} B"


def usable_cores():
    """Host cores this process may really use: affinity mask and cgroup CPU quota, not just nproc."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(blocks, method, budget_s):
    """Reference libzpaq on this box's host cores over a bounded sample of the same blocks.
    Returns (json object, reference archives: list with None for blocks not coded, or None)."""
    from oracle.oracle_py import Oracle, Ref, have_ref
    cores = usable_cores()
    nb, bs = blocks.shape
    if have_ref():
        ref = Ref()
        # calibrate on one block, then a work queue over all usable cores, hard-bounded by a deadline
        t1, _ = ref.compress_blocks_mt(blocks[:1], method, 1)
        per_core = max(t1, 1e-3)
        sample = int(max(1, min(nb, (budget_s / per_core) * cores)))
        threads = min(cores, sample)
        wall, lens, archives = ref.compress_blocks_mt(blocks[:sample], method, threads, deadline_s=budget_s, keep=True)
        done = [i for i, v in enumerate(lens) if v >= 0]
        return {"value": len(done) * bs / 1e6 / wall, "unit": "MB/s", "cores": threads, "kind": "reference",
                "sample": f"{len(done)} x {bs} B blocks of the same corpus, " +
                          (f"libzpaq::Compressor::startBlock({method[1:]}) " if method.startswith("L") else f"libzpaq::compressBlock(\"{method}\") ") +
                          f"(reference built {ref.build_flags()} with its x86 JIT) from a {threads}-thread work queue "
                          f"(nproc={os.cpu_count()}, usable={cores}: the box's CPU quota, a full host would be about "
                          f"{os.cpu_count() / max(cores, 1):.0f}x this); 1 thread alone: {bs / 1e6 / t1:.3f} MB/s",
                "nproc": os.cpu_count(), "usable_cores": cores,
                "single_thread_MBps": bs / 1e6 / t1}, archives
    # fallback: our scalar C port, single thread, a slice of one block
    import zpaq_amd as z
    orc = Oracle()
    k = min(bs, 200000)
    h, _, _ = z.method_to_header(z.expand_method(method, blocks[0][:k]))
    t0 = time.time()
    orc.encode(h, b"\0" + blocks[0][:k].tobytes())
    wall = time.time() - t0
    return {"value": k / 1e6 / wall, "unit": "MB/s", "cores": 1, "kind": "port",
            "sample": f"first {k} B of block 0 through oracle/zpaq_oracle.c (scalar, 1 thread)"}, None


def cpu_decode_baseline(blocks, method, budget_s):
    """Reference libzpaq::decompress (Decompresser, libzpaq.cpp:2247-2374) on this box's host cores over a bounded sample:
    the archives are made by THIS library (bit-identical to the reference's, which the encode leg checks), decoded by the
    reference from a work queue (a decode error fails the leg)."""
    from oracle.oracle_py import Ref, have_ref
    import zpaq_amd as z
    if not have_ref():
        return None
    ref = Ref()
    cores = usable_cores()
    nb, bs = blocks.shape
    one = z.compress_blocks([blocks[0]], method)
    t1 = ref.decompress_blocks_mt(one, bs, 1)
    sample = int(max(1, min(nb, (budget_s / max(t1, 1e-3)) * cores)))
    threads = min(cores, sample)
    archives = z.compress_blocks([blocks[i] for i in range(sample)], method)
    wall = ref.decompress_blocks_mt(archives, bs, threads)
    return {"value": sample * bs / 1e6 / wall, "unit": "MB/s", "cores": threads, "kind": "reference",
            "sample": f"{sample} x {bs} B blocks of the same corpus (archives made by this library), libzpaq::decompress "
                      f"(reference built {ref.build_flags()} with its x86 JIT) from a {threads}-thread work queue "
                      f"(nproc={os.cpu_count()}, usable={cores}); 1 thread alone: {bs / 1e6 / t1:.3f} MB/s",
            "nproc": os.cpu_count(), "usable_cores": cores, "single_thread_MBps": bs / 1e6 / t1}


def one_gpu_same_workload(kind, api=False):
    """The committed one-GPU line of the same corpus (profiles/r06_bench_{default,mixed}.json, falling back to round 5's): the
    device-resident value, or with api=True the host-buffer figure of its `api` leg."""
    name = {"text": "default", "mixed": "mixed"}.get(kind)
    if not name:
        return None
    for rnd in ("r06", "r05"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_bench_{name}.json")
        try:
            one = json.load(open(path))
            v = (one.get("api") or {}).get("value") if api else one["value"]
            if v:
                return {"value": v, "unit": "MB/s", "source": f"profiles/{rnd}_bench_{name}.json" + (" (api leg: host buffers)" if api else ""),
                        "workload": one["config"]["workload"]}
        except Exception:
            continue
    return None


def in_library_bench(a, torch, z):
    """--in-library: N GPUs driven by ONE process through the library's own engines (zpq_init(-1): one engine and one
    host thread per device, the batch sharded contiguously by zpq_shard_range) -- the path a multi-threaded libzpaq
    caller gets without torch.distributed.  The boundary is host buffers, so the figure is PCIe-inclusive."""
    from zpaq_amd import corpus, corpus_torch
    have = torch.cuda.device_count()
    devs = os.environ.get("ZPAQ_AMD_DEVICES")
    if not devs:
        if have < a.gpus:
            sys.exit(f"bench.py: --gpus {a.gpus} requested but only {have} GPU(s) are visible on this box")
        devs = os.environ["ZPAQ_AMD_DEVICES"] = ",".join(str(i) for i in range(a.gpus))
    ndev = len([x for x in devs.split(",") if x.strip() != ""])
    z.init(-1)
    z.set_kernel(a.kernel)
    total_blocks = a.blocks * ndev if a.scaling == "weak" else a.blocks
    bs = a.block_bytes
    dev0 = torch.device("cuda", 0)
    if a.kind == "text":
        host = np.empty((total_blocks, bs), np.uint8)
        for b0 in range(0, total_blocks, 1024):
            k = min(1024, total_blocks - b0)
            host[b0:b0 + k] = corpus_torch.text_blocks(k, bs, corpus.BASE_SEED + b0, dev0).cpu().numpy()
        torch.cuda.empty_cache()
    else:
        host = make_corpus(a.kind, total_blocks, bs, first=0, dev=dev0)
    rows = [host[i] for i in range(total_blocks)]
    for _ in range(a.warmup):
        z.compress_blocks(rows, a.method)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        archives = z.compress_blocks(rows, a.method)
    wall = time.perf_counter() - t0
    plan = z.Plan(z.method_to_header(z.expand_method(a.method, host[0]))[0])
    algo = plan.algo_bytes_per_byte * float(total_blocks) * (bs + 1)
    value = float(total_blocks) * bs * a.steps / 1e6 / wall
    back = z.decompress(archives[0], bs + 64) == host[0].tobytes()
    line = {"metric": f"compress MB/s + bit-identical ratio, -m{a.method} over {a.blocks}x{_size_name(bs)} blocks",
            "value": value, "unit": "MB/s", "n_gpus": ndev, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": wall * 1e3 / max(a.steps, 1), "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": f"method \"{a.method}\" x {total_blocks} blocks x {bs} B '{a.kind}' through zpq_compress_blocks on HOST "
                                   f"buffers, one process, in-library engines on devices {devs} (PCIe-inclusive)",
                       "blocks_total": total_blocks, "block_bytes": bs, "corpus": a.kind, "method": a.method,
                       "parallelism": f"in-library engines x{ndev}", "devices": devs},
            "ratio": sum(len(x) for x in archives) / (float(total_blocks) * bs), "all_status_ok": True,
            "roundtrip_verified_blocks": int(back),
            "roofline": {"bound": "hbm", "achieved": algo * a.steps / 1e9 / wall, "peak": HBM_PEAK_GBS * ndev, "unit": "GB/s",
                         "frac": algo * a.steps / 1e9 / wall / (HBM_PEAK_GBS * ndev), "traffic": None,
                         "kernel": "whole zpq_compress_blocks call (wall clock, includes staging and PCIe)"},
            "cpu_baseline": None}
    # like the rank path: what moved the blocks (nothing but the engines' own host-to-device copies, inside the timed call:
    # the batch is cut by zpq_shard_range, no collective), and the one-GPU figure of the SAME workload through the same entry
    # point (the `api` leg of the committed one-GPU line) for whoever computes an efficiency
    ph = (C.c_double * 8)()
    z.lib().zpq_last_api_timing(ph)
    line["dist_ms"] = {"scatter": 0.0, "gather": 0.0, "scatter_bytes": 0, "gather_bytes": 0,
                       "what": "no collective and no device-to-device traffic: host buffers sharded contiguously over the engines, each engine's "
                               "H2D / D2H copies are part of the timed call",
                       "last_call_ms": {"library_total": ph[0], "host_front": ph[1], "device_call": ph[2], "stitch": ph[3]}}
    if ndev > 1:
        one = one_gpu_same_workload(a.kind, api=True)
        if one:
            line["one_gpu_same_workload"] = one
    if a.cpu_seconds > 0:
        base, _ = cpu_baseline(host, a.method, a.cpu_seconds)
        line["cpu_baseline"] = base
        line["vs_cpu"] = value / base["value"] if base["value"] else None
    print(json.dumps(line))



def configs1_leg(cpu_seconds, timeout_s=150.0):
    """BASELINE configs[1] beside the headline: method "3" (LZ77 through the suffix array + the n = 2 chain ICM, ISSE) over
    256 x 256 KiB LCG blocks, by THIS script in a child process (its own engine; what this process holds stays where it is):
    device-resident `value`, the end-to-end `api` figure, roofline, the reference on the host cores.  A failing child leaves
    an {"error": ...} object: the headline line must survive its side legs."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--method", "3", "--kind", "lcg", "--blocks", "256", "--block-bytes", "262144",
           "--decode-blocks", "0", "--configs1", "0", "--cpu-seconds", str(cpu_seconds)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode or not lines:
            return {"error": ("rc %d: " % r.returncode) + (r.stderr or r.stdout)[-400:]}
        d = json.loads(lines[-1])
        keep = ("metric", "value", "unit", "ms_per_step", "all_status_ok", "roundtrip_verified_blocks", "kernel_ms",
                "persistent_launch", "ratio", "vs_cpu")
        obj = {k: d.get(k) for k in keep}
        obj["config"] = {"workload": (d.get("config") or {}).get("workload")}
        rf = d.get("roofline") or {}
        obj["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_s_per_launch", "algo_bytes_per_launch")}
        api = d.get("api") or {}
        obj["api"] = {"value": api.get("value"), "unit": api.get("unit"), "ms": api.get("ms")}
        cb = d.get("cpu_baseline")
        obj["cpu_baseline"] = ({k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "bit_identical_vs_reference", "compared_blocks")}
                               if isinstance(cb, dict) else None)
        return obj
    except subprocess.TimeoutExpired:
        return {"error": "child did not finish in %.0f s" % timeout_s}
    except Exception as e:
        return {"error": str(e)[:400]}


def legacy_leg(level, cpu_seconds, timeout_s=240.0):
    """SURVEY 8(d) C2 / C3 beside the headline: the reference's LEGACY built-in models at BASELINE scale -- level 2 (mid.cfg, n = 8)
    over configs[1]'s 256 x 256 KiB LCG blocks, level 3 (max.cfg, n = 22; ~246 MB of state per block: 1024 blocks do not fit the HBM
    together, the engine codes them in residency rounds) over configs[2]'s 1024 x 1 MiB text blocks -- by THIS script in a child
    process: device-resident `value`, roofline, every coded payload against the reference's startBlock(level) output (frozen in
    tests/golden/legacy_sha1.json), every block decoded back on the device, the reference on the host cores."""
    import subprocess
    shape = {2: ["--kind", "lcg", "--blocks", "256", "--block-bytes", "262144"],
             3: ["--kind", "text", "--blocks", "1024", "--block-bytes", "1048576"]}[level]
    cmd = [sys.executable, os.path.abspath(__file__), "--legacy-level", str(level)] + shape + \
          ["--decode-blocks", "0", "--configs1", "0", "--legacy", "0", "--cpu-seconds", str(cpu_seconds)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode or not lines:
            return {"error": ("rc %d: " % r.returncode) + (r.stderr or r.stdout)[-400:]}
        d = json.loads(lines[-1])
        keep = ("metric", "value", "unit", "ms_per_step", "all_status_ok", "roundtrip_verified_blocks", "kernel_ms",
                "persistent_launch", "ratio", "vs_cpu", "reference_identity")
        obj = {k: d.get(k) for k in keep}
        cfg = d.get("config") or {}
        obj["config"] = {"workload": cfg.get("workload"), "ncomp": cfg.get("ncomp"), "state_GiB_per_gpu": cfg.get("state_GiB_per_gpu")}
        rf = d.get("roofline") or {}
        obj["roofline"] = {k: rf.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_s_per_launch", "algo_bytes_per_launch")}
        cb = d.get("cpu_baseline")
        obj["cpu_baseline"] = ({k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "bit_identical_vs_reference", "compared_blocks")}
                               if isinstance(cb, dict) else None)
        return obj
    except subprocess.TimeoutExpired:
        return {"error": "child did not finish in %.0f s" % timeout_s}
    except Exception as e:
        return {"error": str(e)[:400]}


def dry_run_bench(a):
    """--dry-run: the N-rank flow of `bench.py --gpus N` WITHOUT GPUs -- torch.distributed over gloo on 127.0.0.1, CPU tensors,
    a method that has no model (LZ77 on the host: the library needs no device for it).  What it exercises is everything
    around the hot path that a multi-GPU run adds: launching the ranks, the corpus made on rank 0 and scattered in
    contiguous ranges (zpaq_amd.dist.scatter_blocks), every rank coding its range with no collective, the timed region
    between barriers with the maximum over the ranks, the archives gathered to rank 0 in block order, ONE JSON line with
    dist_ms.  tests/test_dist.py runs it with two ranks; the numbers mean nothing."""
    import subprocess
    import torch
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    import torch.distributed as dist
    import zpaq_amd as z
    from zpaq_amd import dist as zd
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if world > 1:
        dist.init_process_group("gloo")
    method = a.method if a.method in ("0", "1", "2") else "1"
    per, bs = (a.blocks if a.blocks and a.blocks <= 64 else 12), min(a.block_bytes, 1 << 16)
    total = per * world + (1 if world > 1 else 0)          # (an uneven split)
    full = make_corpus("mixed", total, bs, first=0, dev=None) if rank == 0 else None
    dist_ms = {}
    zd.barrier(); t0 = time.perf_counter()
    mine = zd.scatter_blocks(full, total, bs).numpy()
    zd.barrier()
    dist_ms["scatter"] = (time.perf_counter() - t0) * 1e3
    dist_ms["scatter_bytes"] = total * bs
    rows = [mine[i] for i in range(mine.shape[0])]
    for _ in range(a.warmup):
        z.compress_blocks(rows, method)
    zd.barrier(); t0 = time.perf_counter()
    for _ in range(a.steps):
        archives = z.compress_blocks(rows, method)
    zd.barrier()
    elapsed = zd.max_over_ranks(time.perf_counter() - t0)
    zd.barrier(); t0 = time.perf_counter()
    allc = zd.gather_archives(archives)
    zd.barrier()
    dist_ms["gather"] = (time.perf_counter() - t0) * 1e3
    if rank == 0:
        dist_ms["gather_bytes"] = sum(len(x) for x in allc)
        ok = len(allc) == total and z.decompress(b"".join(allc)) == full.tobytes()
        print(json.dumps({"metric": f"compress MB/s, dry run of the {world}-rank flow on CPU (gloo, method {method}: no model, no GPU)",
                          "value": total * bs * a.steps / 1e6 / elapsed, "unit": "MB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": elapsed * 1e3 / max(a.steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "u8", "data": "synthetic", "dry_run": True, "backend": "gloo",
                          "config": {"workload": f"method \"{method}\" x {total} blocks x {bs} B 'mixed', {world} CPU ranks", "blocks_total": total,
                                     "block_bytes": bs, "parallelism": f"blocks/{world}ranks"},
                          "all_status_ok": bool(ok), "archives_in_block_order_and_round_trip": bool(ok), "dist_ms": dist_ms,
                          "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def machine_code_sha256(path):
    """sha256 over the .text and .rodata bytes of the gfx950 code object inside a clang offload bundle (or a bare ELF):
    what the GPU executes, without the bundle's ids and the notes that change with the name of the source file."""
    b = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    if b.startswith(magic):
        n = struct.unpack_from("<Q", b, len(magic))[0]
        p = len(magic) + 8
        elf = None
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", b, p)
            triple = b[p + 24:p + 24 + tl]
            p += 24 + tl
            if b"amdgcn" in triple:
                elf = b[off:off + size]
        if elf is None:
            return None
        b = elf
    if b[:4] != b"\x7fELF":
        return None
    shoff = struct.unpack_from("<Q", b, 0x28)[0]
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
    def sh(i):
        name, typ, flags, addr, off, size = struct.unpack_from("<IIQQQQ", b, shoff + i * shentsize)
        return name, off, size
    _, stroff, strsize = sh(shstrndx)
    strtab = b[stroff:stroff + strsize]
    h = hashlib.sha256()
    for i in range(shnum):
        name, off, size = sh(i)
        nm = strtab[name:strtab.index(b"\0", name)]
        if nm in (b".text", b".rodata"):
            h.update(nm); h.update(b[off:off + size])
    return h.hexdigest()

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=None,
                    help="blocks per GPU (weak scaling) or in total (--scaling strong); default 1024 (BASELINE configs[2]), and "
                         "2048 with --mode decode: configs[4] decodes the 8192-block archive of the 8-GPU run on ONE GPU, "
                         "which holds 2048 blocks of model state at a time -- one such residency wave is a step")
    ap.add_argument("--block-bytes", type=int, default=1 << 20)
    ap.add_argument("--kind", default=None,
                    help="text | lcg | zeros | records | pattern | mixed; default: text on one GPU (BASELINE configs[2]), "
                         "mixed on several (configs[3]: text, text, LCG, records by block index mod 4)")
    ap.add_argument("--method", default="5")
    ap.add_argument("--legacy-level", type=int, default=0,
                    help="1 / 2 / 3: code every block with the reference's built-in model of that level (Compressor::startBlock(int): "
                         "min.cfg / mid.cfg / max.cfg, libzpaq.cpp:2793-2839) instead of a compressBlock method -- SURVEY 8(d) C2 / C3: "
                         "level 2 on configs[1]'s data, level 3 on configs[2]'s")
    ap.add_argument("--legacy", type=int, default=None,
                    help="1: both legacy configurations (mid.cfg over 256 x 256 KiB LCG, max.cfg over 1024 x 1 MiB text) as side objects "
                         "`legacy2` / `legacy3` of the line, each run by a child process; default: on for the default headline run on one GPU")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--cpu-seconds", type=float, default=None,
                    help="wall budget of the cpu_baseline leg (0 = skip; default 15 on one GPU, 0 on several: the contract asks for the "
                         "CPU baseline on rank 0 at N = 1 only)")
    ap.add_argument("--api-blocks", type=int, default=-1,
                    help="blocks of the end-to-end API leg on host buffers (-1 = the whole batch, 0 = skip)")
    ap.add_argument("--verify-blocks", type=int, default=-1, help="blocks decoded back on the device (-1: every block of the batch)")
    ap.add_argument("--verify-bytes", type=int, default=0, help="how much of each of them (0: the whole block)")
    ap.add_argument("--kernel", type=int, default=0, help="zpq_set_kernel: 0 the engine's choice, 3 / 5 with --mode decode: one / two blocks per wavefront")
    ap.add_argument("--decode-blocks", type=int, default=2048,
                    help="blocks of the `decode` leg of an encode run on one GPU (BASELINE configs[4]'s operating point: one "
                         "residency wave of the 8192-block archive = 2048 x 1 MiB), outside the timed region; 0 = skip")
    ap.add_argument("--configs1", type=int, default=None,
                    help="1: BASELINE configs[1] (-m3 over 256 x 256 KiB LCG blocks) as a side object of the line, run by a child "
                         "process after the other legs; default: on for the default headline run on one GPU, off otherwise")
    ap.add_argument("--decode-kind", default=None,
                    help="corpus of the decode leg: default 'mixed' (BASELINE configs[4] decodes configs[3]'s archive); the same as --kind reuses the timed run's payloads")
    ap.add_argument("--mode", choices=["encode", "decode"], default="encode",
                    help="decode = BASELINE configs[4]: time Decoder::decompress over the archive just produced")
    ap.add_argument("--distribute", dest="distribute", action="store_true", default=None,
                    help="N>1 (default there): rank 0 generates the whole corpus and scatters it over RCCL; coded blocks are "
                         "gathered back (timed separately as dist_ms; the hot path itself has no collective)")
    ap.add_argument("--no-distribute", dest="distribute", action="store_false",
                    help="N>1: every rank generates its own blocks (no RCCL traffic but the barrier and the timing reduction)")
    ap.add_argument("--in-library", action="store_true",
                    help="N>1 in ONE process: zpq_init(-1), one engine per device inside the library, the host-buffer batch "
                         "sharded over them (no torch.distributed)")
    ap.add_argument("--dry-run", action="store_true",
                    help="the N-rank flow on CPU: gloo, a method without a model (scatter, code, gather, one line with dist_ms); no GPU needed")
    a = ap.parse_args()
    if a.cpu_seconds is None:
        a.cpu_seconds = 15.0 if a.gpus == 1 else 0.0
    if a.dry_run:
        return dry_run_bench(a)
    if a.legacy_level:
        if a.legacy_level not in (1, 2, 3):
            sys.exit("bench.py: --legacy-level takes 1, 2 or 3")
        a.method = "L%d" % a.legacy_level       # (the label used below; the reference shim understands it as startBlock(level))
        a.api_blocks = 0                        # no batch entry point takes a level: Compressor::startBlock(int) is per block
    if a.blocks is None:
        a.blocks = 2048 if a.mode == "decode" else 1024
    if a.kind is None:
        a.kind = "text" if a.gpus == 1 else "mixed"

    import torch

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.in_library:
        # `python bench.py --gpus N` by itself: one rank per GPU under torch.distributed.run (RCCL over xGMI)
        have = torch.cuda.device_count()
        if have < a.gpus:
            sys.exit(f"bench.py: --gpus {a.gpus} requested but only {have} GPU(s) are visible on this box")
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    import zpaq_amd as z

    if a.in_library:
        return in_library_bench(a, torch, z)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch N ranks (python -m torch.distributed.run "
                 f"--nproc-per-node N bench.py --gpus N ...) or run `python bench.py --gpus N`, which does that itself")
    if torch.cuda.device_count() <= local:
        sys.exit(f"bench.py: rank {rank} wants GPU {local} but only {torch.cuda.device_count()} GPU(s) are visible")
    if a.distribute is None:
        a.distribute = world > 1
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    z.init(local)
    z.set_kernel(a.kernel)

    bs = a.block_bytes
    if a.scaling == "strong":
        lo, hi = a.blocks * rank // world, a.blocks * (rank + 1) // world   # contiguous ranges (zpaq_amd.dist.shard_range)
        nb, first = hi - lo, lo
        total_blocks = a.blocks
    else:
        nb, first = a.blocks, rank * a.blocks
        total_blocks = a.blocks * world
    dist_ms = {}
    if a.distribute and world > 1:
        from zpaq_amd import dist as zd
        full = None
        if rank == 0 and a.kind == "text":        # generated on rank 0's GPU, scattered device to device
            from zpaq_amd import corpus, corpus_torch
            full = corpus_torch.text_blocks(total_blocks, bs, corpus.BASE_SEED, dev)
        elif rank == 0:
            full = make_corpus(a.kind, total_blocks, bs, first=0, dev=dev)
        zd.barrier(); torch.cuda.synchronize(); t0 = time.perf_counter()
        mine = zd.scatter_blocks(full, total_blocks, bs)
        torch.cuda.synchronize(); zd.barrier()
        dist_ms["scatter"] = (time.perf_counter() - t0) * 1e3
        dist_ms["scatter_bytes"] = int(total_blocks) * bs
        blocks = mine.cpu().numpy()
        nb = blocks.shape[0]
        del full, mine
        torch.cuda.empty_cache()
    elif a.kind == "text":
        # same bytes as corpus.zipf_text, generated on this rank's GPU (seconds instead of minutes of host time)
        from zpaq_amd import corpus, corpus_torch
        d_blocks = corpus_torch.text_blocks(nb, bs, corpus.BASE_SEED + first, dev)
        blocks = d_blocks.cpu().numpy()
        del d_blocks
        torch.cuda.empty_cache()
    else:
        blocks = make_corpus(a.kind, nb, bs, first=first, dev=dev)

    # What the coder sees per block: the PP header (0, or 1 + the PCOMP program of a pre-processing method) followed by
    # the block -- or by its LZ77 / BWT stream, made on the host like compressBlock does (host/preproc.cpp).  One plan
    # per distinct header; the headline text corpus has exactly one.
    L = z.lib()
    L.zpq_preprocess_block.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]

    from concurrent.futures import ThreadPoolExecutor

    def coder_input_of(arr):
        def one(b):
            if a.legacy_level:      # Compressor::postProcess(NULL): a 0 byte, then the data, through the built-in model
                return z.builtin_model_header(a.legacy_level), b"\x00", arr[b]
            xm = z.expand_method(a.method, arr[b])
            h, pc, _ = z.method_to_header(xm)
            pp = (b"\x01" + pc) if pc else b"\x00"
            m = re.match(r"[xs](\d+)[,.](\d+)", xm)
            if m and int(m.group(2)) != 0:
                buf = np.array(arr[b], dtype=np.uint8, copy=True)
                out = np.empty(buf.size + buf.size // 2 + 4096, np.uint8)
                ln = C.c_size_t(0)
                if L.zpq_preprocess_block(xm.encode(), buf.ctypes.data, buf.size, out.ctypes.data, out.size, C.byref(ln)):
                    raise RuntimeError(L.zpq_last_error().decode())
                return h, pp, out[:ln.value]
            return h, pp, arr[b]
        with ThreadPoolExecutor(max_workers=min(32, usable_cores())) as ex:
            return list(ex.map(one, range(len(arr))))

    prepared = coder_input_of(blocks)
    headers = {}
    for b, (h, pp, stream) in enumerate(prepared):
        headers.setdefault(h, []).append(b)
    plan_cache = {h: z.Plan(h) for h in headers}
    groups = [(plan_cache[h], idx) for h, idx in headers.items()]
    in_len = [len(pp) + len(stream) for _, pp, stream in prepared]
    algo_bytes = sum(pl.algo_bytes_per_byte * sum(in_len[i] for i in idx) for pl, idx in groups)
    state_bytes = sum(pl.state_bytes * len(idx) for pl, idx in groups)

    # inputs resident in HBM before the timed region: row b = PP header | stream of block b
    stride_in = (max(in_len) + 255) // 256 * 256
    cap = max(in_len) + max(in_len) // 4 + 4096
    stride_out = (cap + 255) // 256 * 256

    def rows_of(prep, stride):
        host = np.zeros((len(prep), stride), np.uint8)
        for b, (_, pp, stream) in enumerate(prep):
            host[b, :len(pp)] = np.frombuffer(pp, np.uint8)
            host[b, len(pp):len(pp) + len(stream)] = stream
        return host

    host_in = rows_of(prepared, stride_in)
    del prepared
    d_in = torch.from_numpy(host_in).to(dev)
    del host_in
    d_out = torch.empty((nb, stride_out), dtype=torch.uint8, device=dev)
    d_res = torch.zeros((nb, 4), dtype=torch.int32, device=dev)

    # one plan pointer per block: the engine groups blocks by plan and runs the groups concurrently
    plan_of = [None] * nb
    for pl, idx in groups:
        for i in idx:
            plan_of[i] = pl
    PA = (C.c_void_p * nb)(*[p._h for p in plan_of])
    IO = (C.c_uint64 * nb)(*[i * stride_in for i in range(nb)])
    IL = (C.c_uint32 * nb)(*in_len)
    OO = (C.c_uint64 * nb)(*[i * stride_out for i in range(nb)])
    OC = (C.c_uint32 * nb)(*[cap] * nb)
    L.zpq_code_device_multi.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_int]

    def step():
        rc = L.zpq_code_device_multi(0, PA, C.c_void_p(d_in.data_ptr()), IO, IL, nb, C.c_void_p(d_out.data_ptr()),
                                     OO, OC, C.c_void_p(d_res.data_ptr()), None, 1)
        if rc:
            raise RuntimeError(L.zpq_last_error().decode())
        t = z.last_timing()
        return t[0], t[1]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()


    def decode_leg(nd, kind):
        """BASELINE configs[4] beside the headline: Decoder::decompress (libzpaq.cpp:2104-2155) over nd blocks in ONE launch --
        by default blocks of configs[3]'s MIXED corpus (text, text, LCG, records: the archive configs[4] decodes), coded here by
        the device encoder; with --decode-kind equal to --kind the coded payloads of the timed run are reused --, every decoded
        byte compared with the coder's input on the device; its own roofline and the reference's Decompresser beside it."""
        reuse = kind == a.kind
        extra = max(nd - nb, 0) if reuse else nd
        nd = (nb + extra if extra else min(nd, nb)) if reuse else nd
        if reuse:
            ins, lens_in, plans = [d_in[:nd]], list(in_len[:nd]), list(plan_of[:nd])
            codes, lens_out = [d_out[:nd]], [int(x) for x in out_len[:nd]]
        else:
            ins, lens_in, plans, codes, lens_out = [], [], [], [], []
        dec_blocks = blocks if reuse else None          # (host copies for the CPU baseline's sample)
        from zpaq_amd import corpus, corpus_torch
        done = 0
        while done < extra:
            part = min(1024, extra - done)              # (a batch the encoder holds in one residency round)
            f0 = corpus.BASE_SEED + first + (nb + done if reuse else done)
            if kind == "text":
                more = corpus_torch.text_blocks(part, bs, f0, dev).cpu().numpy()
            else:
                more = make_corpus(kind, part, bs, first=(first + nb + done if reuse else first + done), dev=dev)
            if dec_blocks is None:
                dec_blocks = more
            prep = coder_input_of(more)
            del more
            for h, _, _ in prep:
                if h not in plan_cache:
                    plan_cache[h] = z.Plan(h)
            lens2 = [len(pp) + len(st) for _, pp, st in prep]
            if max(lens2) > stride_in - 8:
                return {"skipped": "a block of the decode leg is longer than the headline's row stride"}
            pl2 = [plan_cache[h] for h, _, _ in prep]
            d_in2 = torch.from_numpy(rows_of(prep, stride_in)).to(dev)
            del prep
            d_out2 = torch.empty((part, stride_out), dtype=torch.uint8, device=dev)
            r1 = torch.zeros((part, 4), dtype=torch.int32, device=dev)
            rc = L.zpq_code_device_multi(0, (C.c_void_p * part)(*[p._h for p in pl2]), C.c_void_p(d_in2.data_ptr()),
                                         (C.c_uint64 * part)(*[i * stride_in for i in range(part)]), (C.c_uint32 * part)(*lens2),
                                         part, C.c_void_p(d_out2.data_ptr()), (C.c_uint64 * part)(*[i * stride_out for i in range(part)]),
                                         (C.c_uint32 * part)(*[cap] * part), C.c_void_p(r1.data_ptr()), None, 1)
            if rc:
                raise RuntimeError(L.zpq_last_error().decode())
            r1h = r1.cpu().numpy()
            if not (r1h[:, 2] == 0).all():
                return {"skipped": "coding the decode leg's blocks failed"}
            ins.append(d_in2); codes.append(d_out2)
            lens_in += lens2; plans += pl2; lens_out += [int(x) for x in r1h[:, 0]]
            done += part
        src = torch.cat(ins) if len(ins) > 1 else ins[0]
        code = torch.cat(codes) if len(codes) > 1 else codes[0].clone()
        del ins, codes
        # the container's 4-zero terminator behind every coded payload
        lt = torch.tensor(lens_out, dtype=torch.int64, device=dev)
        code.scatter_(1, torch.arange(4, device=dev)[None, :] + lt[:, None], torch.zeros((nd, 4), dtype=torch.uint8, device=dev))
        back = torch.empty((nd, stride_in), dtype=torch.uint8, device=dev)
        r2 = torch.zeros((nd, 4), dtype=torch.int32, device=dev)
        args = ((C.c_void_p * nd)(*[p._h for p in plans]), C.c_void_p(code.data_ptr()),
                (C.c_uint64 * nd)(*[i * stride_out for i in range(nd)]), (C.c_uint32 * nd)(*[n + 4 for n in lens_out]), nd,
                C.c_void_p(back.data_ptr()), (C.c_uint64 * nd)(*[i * stride_in for i in range(nd)]),
                (C.c_uint32 * nd)(*[n + 8 for n in lens_in]), C.c_void_p(r2.data_ptr()), None, 1)
        torch.cuda.synchronize()
        td0 = time.perf_counter()
        rc = L.zpq_code_device_multi(1, *args)
        torch.cuda.synchronize()
        wall = time.perf_counter() - td0
        if rc:
            raise RuntimeError(L.zpq_last_error().decode())
        tm = z.last_timing()
        r2h = r2.cpu().numpy()
        good = bool((r2h[:, 2] == 0).all() and (r2h[:, 0] == np.array(lens_in)).all())
        cols = torch.arange(stride_in, device=dev)[None, :] < torch.tensor(lens_in, device=dev)[:, None]
        good = good and bool(((back == src) | ~cols).all())
        del back, code, src
        torch.cuda.empty_cache()
        algo = float(sum(p.algo_bytes_per_byte * n for p, n in zip(plans, lens_in)))
        code_s = tm[1] / 1e3
        note2 = C.create_string_buffer(512)
        kk, by_chain = set(), {}
        for p_ in set(plans):          # which decoder each chain of the batch got
            kk.add(int(L.zpq_plan_kernel_kind4(p_._h, 1, nd, max(lens_in), note2, 512)))
            by_chain[f"n={p_.ncomp}"] = note2.value.decode(errors="replace")
        kk = sorted(kk)
        org = " | ".join(f"{k_}: {v_}" for k_, v_ in sorted(by_chain.items()))
        kn = "zpq_spec_decode"
        for tag, nm in (("zpq_spec_decode3", "zpq_spec_decode3 (row / mixer wavefronts, blocks of a workgroup in lockstep)"),
                        ("zpq_spec_decode2", "zpq_spec_decode2 (two blocks per wavefront)")):
            if tag in org:
                kn = nm
                break
        obj = {"metric": f"decompress MB/s, -m{a.method} over {nd}x{_size_name(bs)} blocks (BASELINE configs[4]: one residency "
                         f"wave of the 8-GPU archive on one GPU)",
               # like the headline: inputs resident in HBM, state buffers in place -- Predictor::init + the decoding launch; the
               # wall time of this first decode call of the process also grows the engine's arena pool from 1024 to nd blocks
               "value": float(nd) * bs / 1e6 / ((tm[0] + tm[1]) / 1e3), "unit": "MB/s", "blocks": nd, "block_bytes": bs, "corpus": kind,
               "ncomp": sorted({p.ncomp for p in set(plans)}),
               "ms": {"init_arena": tm[0], "code": tm[1], "wall_first_call": wall * 1e3},
               "every_byte_verified": good,
               "roofline": {"bound": "hbm", "achieved": algo / 1e9 / code_s if code_s > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": (algo / 1e9 / code_s / HBM_PEAK_GBS) if code_s > 0 else 0.0, "traffic": None,
                            "kernel": kn, "kernel_kind": kk, "kernel_origin": org,
                            "algo_bytes_per_launch": algo, "kernel_s_per_launch": code_s,
                            },
               "cpu_baseline": None}
        # HBM traffic of the decoding launch from the committed counter passes (profiles/traffic.json "decode": FETCH_SIZE + WRITE_SIZE
        # per decoded byte of each chain, counted over the lockstep decoder's single dispatch): filled when every chain of the
        # batch was counted and the lockstep decoder is what ran
        try:
            dj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("decode", {})
            per = {int(k.split("=")[1]): v for k, v in dj.items() if k.startswith("n=")}
            if "zpq_spec_decode3" in org and a.method == "5" and all(p.ncomp in per for p in set(plans)):
                obj["roofline"]["traffic"] = float(sum(per[p.ncomp]["traffic_per_decoded_byte"] * n for p, n in zip(plans, lens_in)))
                obj["roofline"]["traffic_per_decoded_byte_by_chain"] = {f"n={k}": v["traffic_per_decoded_byte"] for k, v in per.items()}
                obj["roofline"]["traffic_source"] = dj.get("source")
        except Exception:
            pass
        if a.cpu_seconds > 0:
            base = cpu_decode_baseline(dec_blocks, a.method, a.cpu_seconds)
            obj["cpu_baseline"] = base
            obj["vs_cpu"] = obj["value"] / base["value"] if base and base["value"] else None
        return obj

    for _ in range(a.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    init_ms = code_ms = 0.0
    for _ in range(a.steps):
        i_ms, c_ms = step()
        init_ms += i_ms
        code_ms += c_ms
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    persistent = bool(L.zpq_last_persistent()) if a.mode == "encode" else None
    res = d_res.cpu().numpy()[:nb]
    out_len, status = res[:, 0].astype(np.int64), res[:, 2]
    dec_info = None
    if a.mode == "decode":
        # append the container's 4-zero terminator to every coded payload, then time the decoder
        lens_t = torch.from_numpy(out_len.astype(np.int64)).to(dev)
        colz = torch.arange(4, device=dev)[None, :] + lens_t[:, None]
        d_out.scatter_(1, colz, torch.zeros((nb, 4), dtype=torch.uint8, device=dev))
        back = torch.empty((nb, stride_in), dtype=torch.uint8, device=dev)
        r2 = torch.zeros((nb, 4), dtype=torch.int32, device=dev)

        DIL = (C.c_uint32 * nb)(*[int(out_len[j]) + 4 for j in range(nb)])
        DOC = (C.c_uint32 * nb)(*[n + 8 for n in in_len])

        def dstep():
            rc = L.zpq_code_device_multi(1, PA, C.c_void_p(d_out.data_ptr()), OO, DIL, nb, C.c_void_p(back.data_ptr()),
                                         IO, DOC, C.c_void_p(r2.data_ptr()), None, 1)
            if rc:
                raise RuntimeError(L.zpq_last_error().decode())
            return z.last_timing()[1]

        sync_all()
        td0 = time.perf_counter()
        dcode_ms = 0.0
        for _ in range(a.steps):
            dcode_ms += dstep()
        sync_all()
        delapsed = time.perf_counter() - td0
        if world > 1:
            from zpaq_amd import dist as zd
            delapsed = zd.max_over_ranks(delapsed)
        r2h = r2.cpu().numpy()
        dec_ok = bool((r2h[:, 2] == 0).all() and (r2h[:, 0] == np.array(in_len)).all())
        cols = torch.arange(stride_in, device=dev)[None, :] < torch.tensor(in_len, device=dev)[:, None]
        dec_ok = dec_ok and bool(((back == d_in) | ~cols).all())
        dec_info = {"elapsed": delapsed, "code_ms": dcode_ms, "ok": dec_ok}
        del back
    if a.distribute and world > 1:
        from zpaq_amd import dist as zd
        host_out = d_out.cpu().numpy()
        mine_coded = [host_out[i, :int(out_len[i])].tobytes() for i in range(nb)]
        zd.barrier(); t0 = time.perf_counter()
        allc = zd.gather_archives(mine_coded)
        zd.barrier()
        dist_ms["gather"] = (time.perf_counter() - t0) * 1e3
        if rank == 0:
            assert len(allc) == total_blocks
            dist_ms["gather_bytes"] = sum(len(x) for x in allc)
        del allc
    ok = bool((status == 0).all())
    coded_total = int(out_len.sum())

    # product-path self check: round-trip a few blocks through the device decoder
    verified = 0
    nv = nb if a.verify_blocks < 0 else min(a.verify_blocks, nb)
    if ok and nv:
        vb = (min(min(in_len[:nv]) - 1, a.verify_bytes) + 1) if a.verify_bytes > 0 else max(in_len[:nv]) + 8      # "decode first k bytes" (Decompresser::decompress(n)) or everything
        coded = d_out[:nv].clone()
        lens = [int(out_len[i]) for i in range(nv)]
        for k, ln in enumerate(lens):           # append the 4-zero terminator the container adds
            coded[k, ln:ln + 4] = 0
        back = torch.empty((nv, stride_in), dtype=torch.uint8, device=dev)
        r2 = torch.zeros((nv, 4), dtype=torch.int32, device=dev)
        vPA = (C.c_void_p * nv)(*[plan_of[i]._h for i in range(nv)])
        vio = (C.c_uint64 * nv)(*[k * stride_out for k in range(nv)])
        vil = (C.c_uint32 * nv)(*[ln + 4 for ln in lens])
        voo = (C.c_uint64 * nv)(*[k * stride_in for k in range(nv)])
        voc = (C.c_uint32 * nv)(*[vb if vb < in_len[k] else in_len[k] + 8 for k in range(nv)])
        rc = L.zpq_code_device_multi(1, vPA, C.c_void_p(coded.data_ptr()), vio, vil, nv, C.c_void_p(back.data_ptr()),
                                     voo, voc, C.c_void_p(r2.data_ptr()), None, 0)
        torch.cuda.synchronize()
        if rc:
            raise RuntimeError(L.zpq_last_error().decode())
        r2h = r2.cpu().numpy()
        want = np.array([vb if vb < in_len[k] else in_len[k] for k in range(nv)], np.int64)
        cols = torch.arange(stride_in, device=dev)[None, :] < torch.from_numpy(want).to(dev)[:, None]
        same_rows = (((back == d_in[:nv]) | ~cols).all(dim=1)).cpu().numpy()
        good_rows = (r2h[:, 2] == 0) & (r2h[:, 0] == want) & same_rows
        verified = int(good_rows.sum())
        ok = ok and bool(good_rows.all())
        del back, coded, cols

    # which kernel coded the blocks (4 pipelined encoder, 3 per-header wavefront kernel, 2 generic wave, 1 generic one-lane)
    note = C.create_string_buffer(512)
    dec = 1 if a.mode == "decode" else 0
    kinds = sorted({int(L.zpq_plan_kernel_kind4(pl._h, dec, len(idx), max(in_len[i] for i in idx), note, 512)) for pl, idx in groups})
    kname = {4: "zpq_pipe_{hcomp,rows,light,icm,isse,mix}: one launch of each per step, concurrent",
             3: "zpq_spec_" + ("decode" if dec else "encode"), 2: "code_wave_kernel", 1: "code_serial_kernel"}.get(kinds[-1], "?")
    origin = note.value.decode(errors="replace")
    if dec and "zpq_spec_decode3" in origin:
        kname = "zpq_spec_decode3 (row / mixer wavefronts, blocks of a workgroup in lockstep)"
    elif dec and "zpq_spec_decode2" in origin:
        kname = "zpq_spec_decode2 (two blocks per wavefront)"
    if not dec and persistent and kinds[-1] == 4:
        kname = "zpq_pipe_persist: ONE launch per chain for the whole sequence (unit wavefronts waiting on progress counters)"
    # HBM traffic per launch from the committed rocprofv3 PMC passes -- only when THIS workload was profiled with THIS
    # code object (the cache key is part of kernel_origin); a stale entry is refused
    traffic = None
    transactions = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        key = f"method {a.method} x {nb} x {bs} {a.kind} {a.mode}"
        entries = tj.get(key, [])
        entries = entries if isinstance(entries, list) else [entries]       # one entry per code object that was profiled
        obj_sha = code_sha = None
        if origin.startswith("cache:"):          # the code object that ran: same machine code as the profiled one also counts
            obj = os.path.join(ROOT, "zpaq_amd", "spec_cache", origin[6:].split()[0] + ".hsaco")
            if os.path.exists(obj):
                obj_sha = hashlib.sha256(open(obj, "rb").read()).hexdigest()
                code_sha = machine_code_sha256(obj)
        hit = [e for e in entries if e.get("kernel_origin") == origin or (obj_sha and e.get("code_object_sha256") == obj_sha)
               or (code_sha and e.get("machine_code_sha256") == code_sha)]
        if hit:
            tj = {key: hit[0]}
            traffic = tj[key]["traffic_bytes"]
            # what really bounds the encoder: random memory TRANSACTIONS.  ONE definition, used everywhere:
            # a transaction = a read request of the L2's memory side = FETCH_SIZE / 64 B (calibrated on profiles/r03/gups.hip:
            # a random 16-byte load counts 64 B).  Nearly every one is the read half of a read-modify-write of a table row, so the
            # ceiling is the rate the same microbenchmark reaches for "16-B load + store back": 20-24 G/s over 1-96 GiB footprints.
            # Writes are not added: WRITE_SIZE tallies 32-byte sectors, which counts a coalesced stream store several times.
            transactions = {"per_input_byte": tj[key]["fetch_bytes_per_input_byte"] / 64.0,
                            "counted_as": "read requests: FETCH_SIZE / 64 B; the write halves of the rows' read-modify-writes and the "
                                          "stream stores are not added (WRITE_SIZE counts sectors, not requests)",
                            "peak_G_per_s": 24.0, "peak_source": "profiles/r03/gups_results.txt: 20-24 G random 16-B read-modify-writes/s (49-58 G/s for loads alone)"}
    except Exception:
        pass
    total_bytes = float(total_blocks) * bs * a.steps
    if dec_info:
        elapsed, code_ms, ok = dec_info["elapsed"], dec_info["code_ms"], ok and dec_info["ok"]
    value = total_bytes / 1e6 / elapsed
    code_s = code_ms / 1e3 / max(a.steps, 1)          # coding time per step (this rank)
    achieved = algo_bytes / 1e9 / code_s if code_s > 0 else 0.0
    # which BASELINE.json configuration this run is, said only when it really is that one
    std = a.method == "5" and bs == (1 << 20) and nb == 1024 and a.scaling == "weak"
    which_config = ""
    if std and world == 1 and a.kind == "text" and a.mode == "encode":
        which_config = " = BASELINE configs[2]"
    elif std and world == 8 and a.kind == "mixed" and a.mode == "encode":
        which_config = " = BASELINE configs[3]"
    elif std and world > 1 and a.kind == "mixed" and a.mode == "encode":
        which_config = f" = BASELINE configs[3]'s corpus and per-GPU load on {world} of its 8 GPUs"
    elif a.mode == "decode" and a.method == "5" and bs == (1 << 20) and nb == 2048 and world == 1:
        which_config = " = one residency wave of BASELINE configs[4]"
    if a.legacy_level == 2 and a.kind == "lcg" and nb == 256 and bs == (1 << 18):
        which_config = " = SURVEY 8(d) C2: BASELINE configs[1]'s data through Compressor::startBlock(2) (mid.cfg)"
    elif a.legacy_level == 3 and a.kind == "text" and nb == 1024 and bs == (1 << 20):
        which_config = " = SURVEY 8(d) C3: BASELINE configs[2]'s data through Compressor::startBlock(3) (max.cfg)"
    elif a.legacy_level:
        which_config = f" through Compressor::startBlock({a.legacy_level})"
    line = {
        # BASELINE.json's metric, spelled with the method / batch shape of THIS run (the default is its -m5, 1024 x 1 MiB)
        "metric": ("compress" if a.mode == "encode" else "decompress") +
                  f" MB/s + bit-identical ratio, " + (f"built-in model {a.legacy_level} ({['min', 'mid', 'max'][a.legacy_level - 1]}.cfg)" if a.legacy_level else f"-m{a.method}") +
                  f" over {a.blocks}x{_size_name(bs)} blocks",
        "value": value, "unit": "MB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": elapsed * 1e3 / max(a.steps, 1), "higher_is_better": True, "scaling": a.scaling,
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": f"method \"{a.method}\" x {nb} blocks x {bs} B '{a.kind}' per GPU, {total_blocks} in total" + which_config,
                   "blocks_per_gpu": nb, "blocks_total": total_blocks, "block_bytes": bs, "corpus": a.kind,
                   "method": a.method, "plans": len(groups), "ncomp": [g[0].ncomp for g in groups],
                   "parallelism": f"blocks/{world}gpu", "state_GiB_per_gpu": state_bytes / 2 ** 30,
                   # code-generation / engine knobs in force (none = the product's defaults)
                   "knobs": {k: v for k, v in sorted(os.environ.items()) if k.startswith("ZPAQ_AMD_")}},
        "value_is": "coding throughput with the inputs resident in HBM when the timed region starts (the bench contract); "
                    "api.value is SURVEY 8(d)'s end-to-end figure through zpq_compress_blocks on HOST buffers (PCIe-inclusive)",
        "ratio": coded_total / (float(nb) * bs) if nb else None,
        "all_status_ok": ok, "roundtrip_verified_blocks": verified,
        "kernel_ms": {"init_arena": init_ms / max(a.steps, 1), "code": code_ms / max(a.steps, 1)},
        "persistent_launch": persistent,
        "dist_ms": dist_ms or None,
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": kname, "kernel_origin": origin,
                     "algo_bytes_per_launch": algo_bytes, "kernel_s_per_launch": code_s,
                     "transactions": (dict(transactions, achieved_G_per_s=transactions["per_input_byte"] * float(nb) * bs / 1e9 / code_s,
                                           frac=transactions["per_input_byte"] * float(nb) * bs / 1e9 / code_s / transactions["peak_G_per_s"])
                                      if transactions and code_s > 0 else None)},
    }
    if world > 1 and a.kind == "mixed" and a.scaling == "weak" and std:
        # The N = 1 line of a scaling series is configs[2] (text corpus), the N > 1 lines are configs[3]'s corpus (mixed: a second,
        # longer chain for a quarter of the blocks): efficiency against the N = 1 line mixes two workloads.  The one-GPU figure
        # of THIS workload, measured with `python bench.py --kind mixed`, is carried along for whoever computes the ratio.
        one = one_gpu_same_workload("mixed")
        if one:
            line["one_gpu_same_workload"] = one
    if rank == 0:
        # coded payloads of the timed run, for the identity check against the reference
        ncmp = min(nb, 512)
        host_out = d_out[:ncmp].cpu().numpy()
        # ... and EVERY block of the headline corpus against what the reference made of it, frozen in tests/golden/headline_sha1.json
        # (tests/golden/make_headline_golden.py: the unmodified libzpaq, run where /root/reference exists; nothing of this library
        # or of the oracle is involved): SHA-1 of the coded payload + terminator of the timed, device-resident run
        golden = None
        gpath = os.path.join(ROOT, "tests", "golden", "headline_sha1.json")
        if a.mode == "encode" and a.kind == "text" and a.method == "5" and bs == (1 << 20) and first == 0 and os.path.exists(gpath):
            gj = json.load(open(gpath))["blocks"]
            ng = min(nb, len(gj))
            bad = []
            for b0 in range(0, ng, 256):
                part = d_out[b0:min(b0 + 256, ng)].cpu().numpy()
                for j in range(part.shape[0]):
                    i = b0 + j
                    n = int(out_len[i])
                    if n != gj[i]["coded_len"] or hashlib.sha1(part[j, :n].tobytes() + b"\0\0\0\0").hexdigest() != gj[i]["payload_sha1"]:
                        bad.append(i)
                del part
            golden = {"blocks_compared": ng, "identical": not bad, "first_mismatches": bad[:8],
                      "what": "SHA-1 of every block's coded payload + terminator (timed device-resident run) against the reference's, "
                              "frozen in tests/golden/headline_sha1.json"}
            line["reference_identity"] = golden
        lpath = os.path.join(ROOT, "tests", "golden", "legacy_sha1.json")
        if a.mode == "encode" and a.legacy_level and first == 0 and os.path.exists(lpath):
            lj = json.load(open(lpath))["levels"].get(str(a.legacy_level))
            if lj and lj["block_bytes"] == bs and lj["corpus"].startswith(f"zpaq_amd.corpus.block('{a.kind}'"):
                gj = lj["blocks"]
                ng = min(nb, len(gj))
                bad = []
                for b0 in range(0, ng, 256):
                    part = d_out[b0:min(b0 + 256, ng)].cpu().numpy()
                    for j in range(part.shape[0]):
                        i = b0 + j
                        n = int(out_len[i])
                        if n != gj[i]["coded_len"] or hashlib.sha1(part[j, :n].tobytes() + b"\0\0\0\0").hexdigest() != gj[i]["payload_sha1"]:
                            bad.append(i)
                    del part
                line["reference_identity"] = {"blocks_compared": ng, "identical": not bad, "first_mismatches": bad[:8],
                                              "what": f"SHA-1 of every block's coded payload + terminator (timed device-resident run) against what the "
                                                      f"reference's Compressor::startBlock({a.legacy_level}) made of the same block, frozen in "
                                                      f"tests/golden/legacy_sha1.json (tests/golden/make_legacy_golden.py)"}
        if a.mode == "encode" and world == 1 and a.decode_blocks > 0 and ok:
            try:
                line["decode"] = decode_leg(a.decode_blocks, a.decode_kind or "mixed")
            except Exception as e:            # the headline line must survive a failing side leg
                line["decode"] = {"error": str(e)[:500]}
        del d_out
        torch.cuda.empty_cache()
        # ---- end-to-end through the drop-in API on host buffers (SURVEY 8(d)'s metric), outside the timed region ----
        api_archives = None
        napi = nb if a.api_blocks < 0 else min(a.api_blocks, nb)
        if napi and a.mode == "encode" and world == 1:
            # twice: the first call of a process also allocates the engine's staging / IO buffers (page-locked host memory,
            # device buffers); a service calls again and again -- the second call is the one reported, the first one beside it
            first_ms = None
            for _ in range(2 if a.warmup > 0 else 1):
                t0 = time.perf_counter()
                api_archives = z.compress_blocks([blocks[i] for i in range(napi)], a.method)
                wall = time.perf_counter() - t0
                ph = (C.c_double * 8)()
                L.zpq_last_api_timing(ph)
                if first_ms is None:
                    first_ms = ph[0]
            if golden is not None:      # the archives the drop-in API returned: whole-archive SHA-1 against the reference's
                gj = json.load(open(gpath))["blocks"]
                na = min(napi, len(gj))
                badw = [i for i in range(na) if len(api_archives[i]) != gj[i]["len"] or hashlib.sha1(api_archives[i]).hexdigest() != gj[i]["sha1"]]
                golden["api_archives_compared"] = na
                golden["api_archives_identical"] = not badw
                golden["identical"] = golden["identical"] and not badw
            L.zpq_last_persist_abort_ms.restype = C.c_double
            line["api"] = {"value": napi * bs / 1e6 / (ph[0] / 1e3) if ph[0] else None, "unit": "MB/s", "blocks": napi,
                           "persistent_launch": bool(L.zpq_last_persistent()), "persistent_launch_given_up_after_ms": float(L.zpq_last_persist_abort_ms()),
                           "what": "zpq_compress_blocks on host buffers: SHA-1 + method expansion + header assembly, "
                                   "staging + H2D, Predictor init + coding kernels, D2H, archive framing",
                           "ms": {"library_total": ph[0], "host_front": ph[1], "device_call": ph[2], "stitch": ph[3],
                                  "kernel_init": ph[4], "kernel_code": ph[5], "python_wall": wall * 1e3,
                                  "library_total_first_call": first_ms}}
        want_c1 = a.configs1 if a.configs1 is not None else int(world == 1 and a.mode == "encode" and a.method == "5" and a.kind == "text"
                                                                and nb == 1024 and bs == (1 << 20))
        want_legacy = a.legacy if a.legacy is not None else int(world == 1 and a.mode == "encode" and a.method == "5" and a.kind == "text"
                                                                 and nb == 1024 and bs == (1 << 20))
        if (want_c1 or want_legacy) and world == 1:
            # the side legs run in child processes with engines of their own: this process is done with the device (max.cfg's
            # 1024 blocks need 235 GiB of model state -- with the headline's 100 GiB still held here the child's budget refuses)
            try:
                del d_in, d_res
            except Exception:
                pass
            torch.cuda.empty_cache()
            z.shutdown()
        if want_c1 and world == 1:
            line["configs1"] = configs1_leg(min(a.cpu_seconds, 6.0))
        if want_legacy and world == 1:
            line["legacy2"] = legacy_leg(2, min(a.cpu_seconds, 4.0))
            line["legacy3"] = legacy_leg(3, min(a.cpu_seconds, 6.0))
        if a.cpu_seconds > 0 and a.mode == "decode" and a.legacy_level:
            line["cpu_baseline"] = None
        elif a.cpu_seconds > 0 and a.mode == "decode":
            base = cpu_decode_baseline(blocks, a.method, a.cpu_seconds)
            line["cpu_baseline"] = base
            line["vs_cpu"] = value / base["value"] if base and base["value"] else None
        elif a.cpu_seconds > 0:
            base, ref_arch = cpu_baseline(blocks, a.method, a.cpu_seconds)
            line["cpu_baseline"] = base
            if ref_arch is not None:
                # every archive the reference produced against ours, by SHA-1: the whole archive where the API leg made
                # one, and the coded payload of the TIMED device-resident run against the reference archive's payload
                from oracle.oracle_py import parse_block
                done = [i for i, x in enumerate(ref_arch) if x is not None]
                whole = payload = 0
                same = True
                for i in done:
                    if api_archives is not None and i < len(api_archives):
                        same = same and hashlib.sha1(ref_arch[i]).digest() == hashlib.sha1(api_archives[i]).digest()
                        whole += 1
                    if i < ncmp:
                        ps = parse_block(ref_arch[i])["payload_start"]
                        n = int(out_len[i])
                        same = same and (hashlib.sha1(ref_arch[i][ps:ps + n + 4]).digest() ==
                                         hashlib.sha1(host_out[i, :n].tobytes() + b"\0\0\0\0").digest())
                        payload += 1
                line["cpu_baseline"]["bit_identical_vs_reference"] = bool(same)
                line["cpu_baseline"]["compared_blocks"] = len(done)
                line["cpu_baseline"]["compared_how"] = (f"SHA-1 of the whole archive (API leg): {whole} blocks; SHA-1 of the "
                                                        f"coded payload + terminator of the timed run: {payload} blocks")
            line["vs_cpu"] = value / base["value"] if base["value"] else None
        else:
            line["cpu_baseline"] = None
    if world > 1:
        import torch.distributed as dist
        dist.barrier()            # the other ranks wait here while rank 0 times the CPU reference
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
