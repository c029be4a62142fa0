#!/usr/bin/env python3
"""BASELINE.json's north-star sweep in ONE process: method 5 on highly compressible (zeros) and random (LCG) blocks of
64 KiB .. 16 MiB, batches sized to fill the GPU (1024 blocks up to 4 MiB, 512 at 16 MiB: 233 GiB of model state), inputs
resident in HBM, the coding sequence timed with hipEvents -- and the reference libzpaq on this box's host cores beside
every line (a work queue over the usable cores, one block each).

    python profiles/sweep_north.py out.jsonl [--quick]

Each line: MB/s of the device-resident call, roofline fraction on algorithmic bytes, CPU reference MB/s, whether every
status is 0, whether a few blocks decode back (first 32 KiB through the device decoder) and how many of the blocks the
reference coded as well (one per usable core) have the coded payload that sits inside the reference's archive.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lcg_blocks(torch, nb, n, first_seed, dev):
    """[nb, n] uint8 on the device: block b = corpus.lcg_bytes(n, first_seed + b)."""
    from zpaq_amd import corpus_torch
    out = torch.empty((nb, n), dtype=torch.uint8, device=dev)
    step = max(1, (1 << 26) // n)
    for b0 in range(0, nb, step):
        k = min(step, nb - b0)
        seeds = torch.arange(first_seed + b0, first_seed + b0 + k, dtype=torch.int64, device=dev)
        out[b0:b0 + k] = (corpus_torch._lcg_u32(n, seeds) >> 24).to(torch.uint8)
    return out


def main():
    import torch
    import zpaq_amd as z
    from bench import usable_cores, HBM_PEAK_GBS
    from oracle.oracle_py import Ref, have_ref, parse_block
    from zpaq_amd import corpus
    out_path = sys.argv[1]
    quick = "--quick" in sys.argv
    dev = torch.device("cuda", 0)
    z.init(0)
    L = z.lib()
    L.zpq_code_device_multi.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_int]
    ref = Ref() if have_ref() else None
    cores = usable_cores()
    sizes = [(64 << 10, 1024), (256 << 10, 1024), (1 << 20, 1024), (4 << 20, 1024), (16 << 20, 512)]
    if quick:
        sizes = [(64 << 10, 256), (1 << 20, 128)]
    lines = []
    with open(out_path, "w") as fh:
        for bs, nb in sizes:
            for kind in ("zeros", "lcg"):
                stride_in = (bs + 1 + 255) // 256 * 256
                cap = bs + 1 + ((bs + 1) // 4 if kind == "lcg" else (bs >> 6)) + 4096
                stride_out = (cap + 255) // 256 * 256
                d_in = torch.zeros((nb, stride_in), dtype=torch.uint8, device=dev)
                if kind == "lcg":
                    d_in[:, 1:bs + 1] = lcg_blocks(torch, nb, bs, corpus.BASE_SEED, dev)
                d_out = torch.empty((nb, stride_out), dtype=torch.uint8, device=dev)
                d_res = torch.zeros((nb, 4), dtype=torch.int32, device=dev)
                sample = d_in[0, 1:bs + 1].cpu().numpy()
                hdr = z.method_to_header(z.expand_method("5", sample))[0]
                plan = z.Plan(hdr)
                # the default budget is 85 % of what is free; the 16 MiB line needs 233 of the 288 GB for model state.  The need
                # is known here, so the line is skipped when it cannot fit and the engine's own check is lifted otherwise
                # (an allocation that fails comes back as ZPQ_E_NOMEM, not as a crash)
                free_b, total_b = torch.cuda.mem_get_info()
                need = plan.state_bytes * nb + (6 << 20) * nb + nb * stride_out + (8 << 30)
                if need > total_b - nb * stride_in:
                    line = {"block_bytes": bs, "blocks": nb, "kind": kind, "skipped": "needs %.0f GiB" % (need / 2 ** 30)}
                    fh.write(json.dumps(line) + "\n"); print(json.dumps(line), flush=True)
                    del d_in, d_out, d_res
                    torch.cuda.empty_cache()
                    continue
                z.set_state_budget(int(total_b * 0.99))
                PA = (C.c_void_p * nb)(*[plan._h] * nb)
                IO = (C.c_uint64 * nb)(*[i * stride_in for i in range(nb)])
                IL = (C.c_uint32 * nb)(*[bs + 1] * nb)
                OO = (C.c_uint64 * nb)(*[i * stride_out for i in range(nb)])
                OC = (C.c_uint32 * nb)(*[cap] * nb)
                line = {"block_bytes": bs, "blocks": nb, "kind": kind, "state_GiB": plan.state_bytes * nb / 2 ** 30}
                try:
                    t0 = time.perf_counter()
                    rc = L.zpq_code_device_multi(0, PA, C.c_void_p(d_in.data_ptr()), IO, IL, nb, C.c_void_p(d_out.data_ptr()),
                                                 OO, OC, C.c_void_p(d_res.data_ptr()), None, 1)
                    torch.cuda.synchronize()
                    wall = time.perf_counter() - t0
                    if rc:
                        raise RuntimeError(L.zpq_last_error().decode())
                    init_ms, code_ms, _ = z.last_timing()
                    res = d_res.cpu().numpy()
                    out_len = res[:, 0].astype(np.int64)
                    ok = bool((res[:, 2] == 0).all())
                    algo = plan.algo_bytes_per_byte * float(nb) * (bs + 1)
                    line.update({"MBps": nb * bs / 1e3 / code_ms, "MBps_with_init": nb * bs / 1e3 / (code_ms + init_ms), "code_ms": code_ms,
                                 "init_ms": init_ms, "wall_s": wall, "ok": ok, "ratio": float(out_len.sum()) / (nb * bs),
                                 "roofline_frac": algo / 1e9 / (code_ms / 1e3) / HBM_PEAK_GBS})
                    # a few blocks back through the device decoder (first 32 KiB: Decompresser::decompress(n))
                    nv = 4
                    vb = 32768
                    coded = d_out[:nv].clone()
                    for k in range(nv):
                        coded[k, int(out_len[k]):int(out_len[k]) + 4] = 0
                    back = torch.empty((nv, stride_in), dtype=torch.uint8, device=dev)
                    r2 = torch.zeros((nv, 4), dtype=torch.int32, device=dev)
                    vPA = (C.c_void_p * nv)(*[plan._h] * nv)
                    vio = (C.c_uint64 * nv)(*[k * stride_out for k in range(nv)])
                    vil = (C.c_uint32 * nv)(*[int(out_len[k]) + 4 for k in range(nv)])
                    voo = (C.c_uint64 * nv)(*[k * stride_in for k in range(nv)])
                    voc = (C.c_uint32 * nv)(*[vb] * nv)
                    rc = L.zpq_code_device_multi(1, vPA, C.c_void_p(coded.data_ptr()), vio, vil, nv, C.c_void_p(back.data_ptr()),
                                                 voo, voc, C.c_void_p(r2.data_ptr()), None, 0)
                    torch.cuda.synchronize()
                    r2h = r2.cpu().numpy()
                    line["decoded_back"] = int(sum(int(r2h[k, 2] == 0 and r2h[k, 0] == vb and bool((back[k, :vb] == d_in[k, :vb]).all()))
                                                   for k in range(nv))) if rc == 0 else 0
                    # coded payloads of the blocks the reference will code as well (one per usable core)
                    payloads = [d_out[k, :int(out_len[k])].cpu().numpy().tobytes() for k in range(min(cores, nb))]
                    del coded, back
                except Exception as ex:
                    line["error"] = str(ex)[:300]
                    payloads = None
                del d_out, d_res
                # the reference on the host cores: one block per thread from a work queue
                if ref is not None:
                    ncpu = min(cores, nb)
                    host = d_in[:ncpu, 1:bs + 1].cpu().numpy()
                    wall_c, lens, arch = ref.compress_blocks_mt(host, "5", ncpu, keep=True)
                    line["cpu_MBps"] = ncpu * bs / 1e6 / wall_c
                    line["cpu_cores"] = ncpu
                    line["cpu_build"] = ref.build_flags()
                    if payloads is not None:
                        same = 0
                        for k in range(ncpu):
                            ps = parse_block(arch[k])["payload_start"]
                            same += arch[k][ps:ps + len(payloads[k]) + 4] == payloads[k] + b"\0\0\0\0"
                        line["blocks_compared_with_reference"] = ncpu
                        line["blocks_identical_to_reference"] = int(same)
                    if "MBps" in line:
                        line["vs_cpu"] = line["MBps"] / line["cpu_MBps"]
                del d_in
                torch.cuda.empty_cache()
                z.set_state_budget(0)
                lines.append(line)
                fh.write(json.dumps(line) + "\n")
                fh.flush()
                print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
