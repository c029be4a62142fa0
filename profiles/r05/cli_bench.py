#!/usr/bin/env python3
"""End to end through the reference's archiver, wall clock, a tree of N files of 1 MiB of the text corpus:
    oracle/_ref/zpaq_ref_cli        the reference as it is (its own libzpaq.cpp, x86 JIT, host cores)
    oracle/_ref/zpaq_amd_cli        the reference's zpaq.cpp, unmodified, on this library (one block per thread, coalesced)
    oracle/_ref/zpaq_amd_cli_batch  the same with patches/zpaq_batch.patch (the job queues handed to the batch API)
`add -method 50` (level 5, 1 MiB blocks) and `extract`; every archive is extracted by another binary and the trees compared.

    python profiles/r05/cli_bench.py [--files 256] [--out gpurun_out/r05/cli.json]
"""
import argparse, filecmp, json, os, shutil, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=256)
    ap.add_argument("--work", default="/tmp/zpq_cli_bench")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05", "cli.json"))
    ap.add_argument("--unpatched-threads", default="64")
    ap.add_argument("--timeout", type=float, default=240.0)
    ap.add_argument("--skip-unpatched-extract", action="store_true", help="the unpatched archiver's extract takes ~85 s of GPU time: leave it out")
    ap.add_argument("--keep", action="store_true", help="leave the work directory (the tree of files) behind")
    ap.add_argument("--quick", action="store_true", help="only the patched archiver (add twice, extract), with the library's one line per device batch (ZPAQ_AMD_LOG)")
    a = ap.parse_args()
    import torch
    from zpaq_amd import corpus, corpus_torch
    ref, ours, batch = (os.path.join(ROOT, "oracle", "_ref", n) for n in ("zpaq_ref_cli", "zpaq_amd_cli", "zpaq_amd_cli_batch"))
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

    def save():
        os.makedirs(os.path.dirname(a.out), exist_ok=True)
        json.dump({"files": a.files, "file_bytes": bs, "total_bytes": total, "nproc": os.cpu_count(), "rows": rows}, open(a.out, "w"), indent=1)

    def run(exe, args, label, threads, check_tree=None):
        t0 = time.perf_counter()
        try:
            env = dict(os.environ, ZPAQ_AMD_LOG="1", ZPAQ_AMD_PERSIST_PROF=os.path.join(os.path.dirname(a.out), "cli_prof_%d.bin" % len(rows))) if a.quick else None
            r = subprocess.run([exe] + args + ["-threads", str(threads)], cwd=a.work, capture_output=True, text=True, timeout=a.timeout, env=env)
        except subprocess.TimeoutExpired:
            rows.append({"what": label, "threads": threads, "timeout_s": a.timeout})
            print(json.dumps(rows[-1]), flush=True); save()
            return False
        wall = time.perf_counter() - t0
        row = {"what": label, "threads": threads, "wall_s": wall, "MBps": total / 1e6 / wall, "rc": r.returncode}
        if r.returncode:
            row["stderr"] = r.stderr[-600:]
        if a.quick:
            row["library_log"] = [l for l in r.stderr.splitlines() if l.startswith("[zpaq_amd]")]
            row["archiver_says"] = [l for l in (r.stdout + r.stderr).splitlines() if "seconds" in l][-2:]
        if check_tree and os.path.isdir(os.path.join(check_tree, "tree")):
            c = filecmp.dircmp(tree, os.path.join(check_tree, "tree"))
            _, mism, errs = filecmp.cmpfiles(tree, os.path.join(check_tree, "tree"), c.common_files, shallow=False)
            row["tree_identical"] = not (c.left_only or c.right_only or mism or errs)
            shutil.rmtree(check_tree, ignore_errors=True)
        rows.append(row)
        print(json.dumps(row), flush=True); save()
        return r.returncode == 0

    A = lambda n: os.path.join(a.work, n)
    run(batch, ["add", A("w.zpaq"), "tree", "-method", "50"], "zpaq_amd_cli_batch add -method 50 (first call of the box: code objects, page-locked buffers)", 4)
    run(batch, ["add", A("batch.zpaq"), "tree", "-method", "50"], "zpaq_amd_cli_batch add -method 50", 4)
    rows[-1]["archive_bytes"] = os.path.getsize(A("batch.zpaq")) if os.path.exists(A("batch.zpaq")) else None
    if a.quick:
        run(batch, ["extract", A("batch.zpaq"), "-to", A("x_batch")], "zpaq_amd_cli_batch extract (its own archive)", 4, A("x_batch"))
        save()
        if not a.keep: shutil.rmtree(a.work, ignore_errors=True)
        return
    run(ref, ["add", A("ref.zpaq"), "tree", "-method", "50"], "zpaq_ref_cli add -method 50", 16)
    rows[-1]["archive_bytes"] = os.path.getsize(A("ref.zpaq")) if os.path.exists(A("ref.zpaq")) else None
    for T in [int(x) for x in a.unpatched_threads.split(",") if x]:
        run(ours, ["add", A(f"ours{T}.zpaq"), "tree", "-method", "50"], "zpaq_amd_cli (unpatched) add -method 50", T)
    run(batch, ["extract", A("ref.zpaq"), "-to", A("x_batch")], "zpaq_amd_cli_batch extract (the reference's archive)", 4, A("x_batch"))
    run(ref, ["extract", A("batch.zpaq"), "-to", A("x_ref")], "zpaq_ref_cli extract (the batch archiver's archive)", 16, A("x_ref"))
    for T in [] if a.skip_unpatched_extract else [int(x) for x in a.unpatched_threads.split(",") if x]:
        run(ours, ["extract", A("ref.zpaq"), "-to", A("x_ours")], "zpaq_amd_cli (unpatched) extract (the reference's archive)", max(T, 256), A("x_ours"))
    save()
    shutil.rmtree(a.work, ignore_errors=True)


if __name__ == "__main__":
    main()
