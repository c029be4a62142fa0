#!/usr/bin/env python3
"""End-to-end through the UNMODIFIED reference archiver:
    oracle/_ref/zpaq_amd_cli = the reference's zpaq.cpp linked against this library (GPU coder behind libzpaq's API)
    oracle/_ref/zpaq_ref_cli = the reference as it is (its own libzpaq.cpp, x86 JIT, host cores)
`add -method 50` (level 5, 1 MiB blocks) and `extract` over a tree of N files of 1 MiB of the text corpus, wall clock.
zpaq.cpp runs compressBlock / Decompresser from a pool of -threads T (zpaq.cpp:1918-1965, 2848-2867): T blocks reach the
library at a time, where the submission queue coalesces them into device batches.

    python profiles/r04/cli_bench.py [--files 1024] [--out gpurun_out/r04/cli.json]
"""
import argparse
import filecmp
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=256)
    ap.add_argument("--work", default="/tmp/zpq_cli_bench")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04", "cli.json"))
    ap.add_argument("--threads", default="16,64,256")
    ap.add_argument("--ref-threads", default="16")
    ap.add_argument("--extract-threads", default="64,256", help="ours: a block decodes at ~65 KB/s whatever the batch holds, so few threads crawl")
    ap.add_argument("--timeout", type=float, default=400.0)
    a = ap.parse_args()
    import torch
    from zpaq_amd import corpus, corpus_torch
    ours, ref = (os.path.join(ROOT, "oracle", "_ref", n) for n in ("zpaq_amd_cli", "zpaq_ref_cli"))
    shutil.rmtree(a.work, ignore_errors=True)
    tree = os.path.join(a.work, "tree")
    os.makedirs(tree)
    dev = torch.device("cuda", 0)
    bs = 1 << 20
    for b0 in range(0, a.files, 256):
        k = min(256, a.files - b0)
        t = corpus_torch.text_blocks(k, bs, corpus.BASE_SEED + b0, dev).cpu().numpy()
        for j in range(k):
            t[j].tofile(os.path.join(tree, f"f{b0 + j:05d}.txt"))
    del t
    torch.cuda.empty_cache()
    total = a.files * bs
    rows = []

    def save():          # after every row: a call that runs out of time keeps what it measured
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump({"files": a.files, "file_bytes": bs, "total_bytes": total, "nproc": os.cpu_count(), "rows": rows}, open(a.out, "w"), indent=1)

    def run(exe, args, label, threads):
        t0 = time.perf_counter()
        try:
            r = subprocess.run([exe] + args + ["-threads", str(threads)], cwd=a.work, capture_output=True, text=True, timeout=a.timeout)
        except subprocess.TimeoutExpired:
            rows.append({"what": label, "threads": threads, "timeout_s": a.timeout})
            print(json.dumps(rows[-1]), flush=True)
            save()
            return False
        wall = time.perf_counter() - t0
        row = {"what": label, "threads": threads, "wall_s": wall, "MBps": total / 1e6 / wall, "rc": r.returncode}
        if r.returncode:
            row["stderr"] = r.stderr[-600:]
        rows.append(row)
        print(json.dumps(row), flush=True)
        save()
        return r.returncode == 0

    # a first small call of ours so that the one-off costs of a process (code objects, page-locked buffers) show up separately
    for T in [int(x) for x in a.threads.split(",")]:
        arc = os.path.join(a.work, f"ours{T}.zpaq")
        run(ours, ["add", arc, "tree", "-method", "50"], "zpaq_amd_cli add -method 50", T)
        rows[-1]["archive_bytes"] = os.path.getsize(arc) if os.path.exists(arc) else None
    for T in [int(x) for x in a.ref_threads.split(",")]:
        arc = os.path.join(a.work, f"ref{T}.zpaq")
        run(ref, ["add", arc, "tree", "-method", "50"], "zpaq_ref_cli add -method 50", T)
        rows[-1]["archive_bytes"] = os.path.getsize(arc) if os.path.exists(arc) else None
    T0 = int(a.threads.split(",")[0])
    R0 = int(a.ref_threads.split(",")[0])
    for T in [int(x) for x in a.extract_threads.split(",")]:
        to = os.path.join(a.work, f"x_ours{T}")
        run(ours, ["extract", os.path.join(a.work, f"ref{R0}.zpaq"), "-to", to], "zpaq_amd_cli extract (the reference's archive)", T)
        if os.path.isdir(os.path.join(to, "tree")):
            c = filecmp.dircmp(tree, os.path.join(to, "tree"))
            _, mism, errs = filecmp.cmpfiles(tree, os.path.join(to, "tree"), c.common_files, shallow=False)
            rows[-1]["tree_identical"] = not (c.left_only or c.right_only or mism or errs)
        shutil.rmtree(to, ignore_errors=True)
    to = os.path.join(a.work, "x_ref")
    run(ref, ["extract", os.path.join(a.work, f"ours{T0}.zpaq"), "-to", to], "zpaq_ref_cli extract (this library's archive)", R0)
    c = filecmp.dircmp(tree, os.path.join(to, "tree"))
    _, mism, errs = filecmp.cmpfiles(tree, os.path.join(to, "tree"), c.common_files, shallow=False)
    rows[-1]["tree_identical"] = not (c.left_only or c.right_only or mism or errs)
    save()
    shutil.rmtree(a.work, ignore_errors=True)


if __name__ == "__main__":
    main()
