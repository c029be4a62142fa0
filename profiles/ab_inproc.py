#!/usr/bin/env python3
"""A/B of code-generation knobs in ONE process: the corpus is generated once and stays in HBM; every variant gets a fresh
plan (the knobs are part of the generated source, so a fresh plan loads that variant's code object), codes the batch
once, and its coded bytes are compared with the first variant's (which the parity tests pin against the reference).

    python profiles/ab_inproc.py --prebuild VARIANTS.json      (no GPU: compiles the variants' code objects into zpaq_amd/spec_cache)
    python profiles/ab_inproc.py VARIANTS.json [--out file]    (on the MI355X)

VARIANTS.json: {"blocks": 1024, "block_bytes": 1048576, "kind": "text", "method": "5", "runs": 1,
                "variants": [["name", {"ZPAQ_AMD_...": "1"}], ...]}
A process per variant (bench.py) spends ~40 s on start-up, corpus and verification; this spends the coding time only.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def knob_env(env):
    for k in [k for k in os.environ if k.startswith("ZPAQ_AMD_PIPE_") or k in ("ZPAQ_AMD_SPEC_DEFS",)]:
        del os.environ[k]
    for k, v in env.items():
        os.environ[k] = str(v)


def header_of(method, block):
    import zpaq_amd as z
    xm = z.expand_method(method, block)
    h, pc, _ = z.method_to_header(xm)
    assert not pc, "A/B harness codes the block itself (no pre-processing methods)"
    return h


def prebuild(cfg):
    import zpaq_amd as z
    from zpaq_amd import corpus, prebuild as pb
    L = z.lib()
    L.zpq_spec_cache_dir.restype = C.c_char_p
    L.zpq_spec_include_dir.restype = C.c_char_p
    cache, inc = L.zpq_spec_cache_dir().decode(), L.zpq_spec_include_dir().decode()
    os.makedirs(cache, exist_ok=True)
    kinds = ["text", "text", "lcg", "records"] if cfg["kind"] == "mixed" else [cfg["kind"]]
    headers = {header_of(cfg["method"], corpus.block(k, cfg["block_bytes"], corpus.BASE_SEED + i)) for i, k in enumerate(kinds)}
    jobs = []
    for name, env in cfg["variants"]:
        knob_env(env)
        for h in headers:
            for variant in (0, 1, 2):                    # the engine picks one by batch size and block length
                src, key = pb.pipe_source_and_key(h, variant)
                if src is None:
                    print(f"{name}: no pipelined encoder ({key})")
                    continue
                jobs.append((src, key, cache, inc, dict(env)))
            for waves in cfg.get("spec_waves", []):          # the per-header wavefront kernel (decoder) as well
                os.environ["ZPAQ_AMD_SPEC_WAVES"] = str(waves)
                src, key = pb.source_and_key(h)
                os.environ.pop("ZPAQ_AMD_SPEC_WAVES", None)
                if src is not None:
                    jobs.append((src, key, cache, inc, dict(env)))
    from concurrent.futures import ThreadPoolExecutor

    def one(j):
        src, key, cache, inc, env = j
        if env.get("ZPAQ_AMD_SPEC_DEFS"):
            os.environ["ZPAQ_AMD_SPEC_DEFS"] = env["ZPAQ_AMD_SPEC_DEFS"]     # (serialised below when any variant sets it)
        else:
            os.environ.pop("ZPAQ_AMD_SPEC_DEFS", None)
        return pb.compile_one((src, key, cache, inc))
    defs = any("ZPAQ_AMD_SPEC_DEFS" in e for _, e in cfg["variants"])
    os.environ.pop("ZPAQ_AMD_SPEC_DEFS", None)
    with ThreadPoolExecutor(max_workers=1 if defs else min(8, os.cpu_count() or 1)) as ex:
        res = list(ex.map(one, jobs))
    print(f"{sum(1 for _, s in res if s == 'built')} built, {sum(1 for _, s in res if s == 'cached')} cached")


def main():
    pre = "--prebuild" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cfg = json.load(open(args[0]))
    out_path = None
    if "--out" in sys.argv:
        out_path = sys.argv[sys.argv.index("--out") + 1]
        args = [a for a in args if a != out_path]
    if pre:
        prebuild(cfg)
        return
    import torch
    import zpaq_amd as z
    from zpaq_amd import corpus, corpus_torch
    dev = torch.device("cuda", 0)
    z.init(0)
    L = z.lib()
    nb, bs, kind, method = cfg["blocks"], cfg["block_bytes"], cfg["kind"], cfg["method"]
    if kind == "text":
        blocks_d = corpus_torch.text_blocks(nb, bs, corpus.BASE_SEED, dev)
        sample = blocks_d[:1].cpu().numpy()
        headers = [header_of(method, sample[0])] * nb
    else:
        sys.path.insert(0, ROOT)
        from bench import make_corpus
        host = make_corpus(kind, nb, bs, first=0)
        headers = [header_of(method, host[b]) for b in range(nb)]
        blocks_d = torch.from_numpy(host).to(dev)
    stride_in = (bs + 1 + 255) // 256 * 256
    d_in = torch.zeros((nb, stride_in), dtype=torch.uint8, device=dev)
    d_in[:, 1:bs + 1] = blocks_d
    del blocks_d
    cap = bs + 1 + (bs + 1) // 4 + 4096
    stride_out = (cap + 255) // 256 * 256
    d_out = torch.empty((nb, stride_out), dtype=torch.uint8, device=dev)
    d_res = torch.zeros((nb, 4), dtype=torch.int32, device=dev)
    IO = (C.c_uint64 * nb)(*[i * stride_in for i in range(nb)])
    IL = (C.c_uint32 * nb)(*[bs + 1] * nb)
    OO = (C.c_uint64 * nb)(*[i * stride_out for i in range(nb)])
    OC = (C.c_uint32 * nb)(*[cap] * nb)
    L.zpq_code_device_multi.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64),
                                        C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_int]
    ref_out = ref_len = None
    lines = []
    for name, env in cfg["variants"]:
        knob_env(env)
        plans = {}
        try:
            for h in set(headers):
                plans[h] = z.Plan(h)
            PA = (C.c_void_p * nb)(*[plans[h]._h for h in headers])
            best = None
            for _ in range(int(cfg.get("runs", 1))):
                d_out.zero_()
                t0 = time.perf_counter()
                rc = L.zpq_code_device_multi(0, PA, C.c_void_p(d_in.data_ptr()), IO, IL, nb, C.c_void_p(d_out.data_ptr()),
                                             OO, OC, C.c_void_p(d_res.data_ptr()), None, 1)
                torch.cuda.synchronize()
                wall = time.perf_counter() - t0
                if rc:
                    raise RuntimeError(L.zpq_last_error().decode())
                code_ms = z.last_timing()[1]
                best = code_ms if best is None else min(best, code_ms)
            res = d_res.cpu().numpy()
            ok = bool((res[:, 2] == 0).all())
            note = C.create_string_buffer(512)
            kd = int(L.zpq_plan_kernel_kind4(next(iter(plans.values()))._h, 0, nb, bs + 1, note, 512))
            if ref_out is None:
                ref_out, ref_len = d_out.clone(), res[:, 0].copy()
                same = True
            else:
                same = bool((res[:, 0] == ref_len).all()) and bool(torch.equal(d_out, ref_out))
            line = {"name": name, "env": env, "code_ms": best, "MBps": nb * bs / 1e3 / best, "status_ok": ok,
                    "same_bytes_as_first": same, "kernel_kind": kd, "origin": note.value.decode(errors="replace")[:60],
                    "wall_s": wall}
        except Exception as ex:     # a variant that does not load / run must not end the sweep
            line = {"name": name, "env": env, "error": str(ex)[:300]}
        lines.append(line)
        print(json.dumps(line), flush=True)
        if out_path:
            with open(out_path, "w") as fh:
                for ln in lines:
                    fh.write(json.dumps(ln) + "\n")
        del plans


if __name__ == "__main__":
    main()
