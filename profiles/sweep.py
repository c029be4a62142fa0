#!/usr/bin/env python3
"""North-star sweep (BASELINE.json): MB/s on highly compressible and random blocks from 64 KiB up,
plus the mixed corpus of config 4 and the decode-only leg of config 5, one MI355X.  Runs bench.py
repeatedly and writes one JSON line per point to the given file.

    python profiles/sweep.py gpurun_out/sweep.jsonl [--max-mib 4]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-seconds", "0", "--warmup", "0", "--api-blocks", "0"] + args
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric')]
    if not lines:
        return {"error": r.stderr[-500:], "args": args}
    j = json.loads(lines[-1])
    return {"args": " ".join(args), "MBps": round(j["value"], 2), "kernel_MBps": round(
        j["config"]["blocks_per_gpu"] * j["config"]["block_bytes"] / 1e3 / j["kernel_ms"]["code"], 2),
        "ratio": j["ratio"], "ok": j["all_status_ok"], "verified": j["roundtrip_verified_blocks"],
        "ncomp": j["config"]["ncomp"], "state_GiB": round(j["config"]["state_GiB_per_gpu"], 1),
        "roofline_frac": round(j["roofline"]["frac"], 4), "kernel": j["roofline"]["kernel"]}


def main():
    out = sys.argv[1]
    max_mib = float(sys.argv[sys.argv.index("--max-mib") + 1]) if "--max-mib" in sys.argv else 4
    pts = []
    which = sys.argv[sys.argv.index("--set") + 1] if "--set" in sys.argv else "all"
    if which in ("all", "sizes"):
        for kind in ("zeros", "lcg"):
            for kib, blocks in ((64, 1024), (256, 1024), (1024, 512), (4096, 256), (16384, 64)):
                if kib / 1024 > max_mib:
                    continue
                pts.append(["--kind", kind, "--blocks", str(blocks), "--block-bytes", str(kib * 1024)])
    if which == "north":      # BASELINE.json north star: random and highly compressible blocks, 64 KiB .. 16 MiB
        for kind in ("zeros", "lcg"):
            for kib, blocks in ((64, 1024), (256, 1024), (1024, 1024), (4096, 256), (16384, 64)):
                pts.append(["--kind", kind, "--blocks", str(blocks), "--block-bytes", str(kib * 1024)])
    if which == "sizes4":
        for kind in ("zeros", "lcg"):
            pts.append(["--kind", kind, "--blocks", "256", "--block-bytes", str(4096 * 1024)])
    if which in ("all", "configs"):
        pts.append(["--kind", "mixed", "--blocks", "1024", "--block-bytes", str(256 * 1024)])      # config 4 shape
        pts.append(["--kind", "text", "--blocks", "1024", "--block-bytes", str(256 * 1024), "--mode", "decode"])   # config 5
        pts.append(["--kind", "text", "--blocks", "1024", "--block-bytes", str(256 * 1024), "--method", "4"])
        pts.append(["--kind", "pattern", "--blocks", "256", "--block-bytes", str(64 * 1024)])     # one chain per block: JIT budget
    if which == "north":
        # known answer of BASELINE.md section 2 at the top of the range: 16 MiB zeros, method 5 -> 688 bytes
        import hashlib
        import time
        import numpy as np
        sys.path.insert(0, ROOT)
        import zpaq_amd as z
        z.init(0)
        t0 = time.time()
        a, = z.compress_blocks([np.zeros(16 << 20, np.uint8)], "5")
        ka = {"known_answer": "16 MiB zeros, method 5", "len": len(a), "sha1": hashlib.sha1(a).hexdigest(),
              "ok": len(a) == 688 and hashlib.sha1(a).hexdigest() == "aecc5f154175bc56a6af2d0015f1e01acef55655",
              "roundtrip": z.decompress(a) == bytes(16 << 20), "seconds": round(time.time() - t0, 1)}
        print(ka, flush=True)
        z.shutdown()
    with open(out, "w") as fh:
        if which == "north":
            fh.write(json.dumps(ka) + "\n")
        for p in pts:
            res = run(p)
            fh.write(json.dumps(res) + "\n")
            fh.flush()
            print(res, flush=True)


if __name__ == "__main__":
    main()
