#!/usr/bin/env python3
"""Every golden archive whose chain the two-blocks-per-wavefront decoder takes, decoded with it on the MI355X
(zpq_set_kernel(5)): the archives were written by the reference (tests/golden/make_golden.py); zpq_decompress checks
their SHA-1 trailers and the bytes are compared with the regenerated input.  Torch-free."""
import base64, json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import zpaq_amd as z
from conftest import gen_input
g = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
z.init(0)
t0 = time.time()
ok = bad = skipped = 0
fails = []
for sect in ("config_cases", "level_cases", "vm_cases", "method_cases"):
    for e in g[sect]:
        if "archive_b64" not in e:
            skipped += 1
            continue
        a = base64.b64decode(e["archive_b64"])
        want = gen_input(e).tobytes()
        hdr = bytes.fromhex(e["header"])
        if hdr[6] == 0 or hdr[6] > 32:
            skipped += 1
            continue
        if time.time() - t0 > 11:
            skipped += 1
            continue
        z.set_kernel(5)
        try:
            got = z.decompress(a, cap=len(want) + 64)
            if got == want:
                ok += 1
            else:
                bad += 1; fails.append((sect, e.get("name") or e.get("method") or e.get("level"), "differs"))
        except Exception as ex:
            msg = str(ex)
            if "unavailable" in msg:
                skipped += 1
            else:
                bad += 1; fails.append((sect, e.get("name") or e.get("method") or e.get("level"), msg[:120]))
z.set_kernel(0)
print(json.dumps({"what": "golden archives through the two-blocks-per-wavefront decoder", "ok": ok, "bad": bad, "skipped": skipped,
                  "fails": fails[:8], "s": round(time.time() - t0, 1)}))
