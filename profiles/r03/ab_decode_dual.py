#!/usr/bin/env python3
"""A/B on the MI355X, torch-free: the decoder with two blocks per wavefront (zpq_set_kernel(5), device/spec_dual_kernel.h)
against one block per wavefront (zpq_set_kernel(3)) on a dense batch of -m5 blocks.  Every decode goes through
zpq_decompress (SHA-1 trailers checked) and is compared with the input.  Prints one JSON line per measurement.

    python profiles/r03/ab_decode_dual.py [blocks] [block_bytes]
"""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import zpaq_amd as z
    from zpaq_amd import corpus
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
    bs = int(sys.argv[2]) if len(sys.argv) > 2 else 128 << 10
    t0 = time.time()
    z.init(0)
    distinct = [corpus.block("text", bs, corpus.BASE_SEED + i).tobytes() for i in range(48)]
    ragged = [corpus.block("text", n, 900 + n).tobytes() for n in (1, 1000, 65537)] + [b""]
    blocks = [distinct[i % len(distinct)] for i in range(nb - len(ragged))] + ragged
    print(json.dumps({"what": "inputs", "blocks": len(blocks), "bytes": sum(map(len, blocks)), "s": round(time.time() - t0, 1)}), flush=True)
    t1 = time.time()
    archives = z.compress_blocks(blocks, "5")
    print(json.dumps({"what": "encode", "s": round(time.time() - t1, 2), "timing": z.last_timing()}), flush=True)
    plain = b"".join(blocks)
    whole = b"".join(archives)
    small = b"".join(archives[-5:])                 # 5 blocks: a wavefront with one block only, the empty block
    small_plain = b"".join(blocks[-5:])
    for kernel, name in ((3, "one block per wavefront"), (5, "two blocks per wavefront"), (3, "one block per wavefront"), (5, "two blocks per wavefront")):
        z.set_kernel(kernel)
        try:
            ok_small = z.decompress(small, cap=len(small_plain) + 16) == small_plain
            t = time.time()
            back = z.decompress(whole, cap=len(plain) + 16)
            wall = time.time() - t
            init_ms, code_ms, n = z.last_timing()
            print(json.dumps({"what": "decode", "kernel": kernel, "name": name, "identical": back == plain, "small_identical": ok_small,
                              "code_ms": round(code_ms, 1), "MBps": round(len(plain) / 1e3 / code_ms, 1), "blocks": n, "wall_s": round(wall, 2)}), flush=True)
        except Exception as ex:
            print(json.dumps({"what": "decode", "kernel": kernel, "error": str(ex)[:400]}), flush=True)
    z.set_kernel(0)
    t = time.time()
    back = z.decompress(whole, cap=len(plain) + 16)
    init_ms, code_ms, n = z.last_timing()
    print(json.dumps({"what": "decode", "kernel": 0, "name": "the engine's choice", "identical": back == plain, "code_ms": round(code_ms, 1),
                      "MBps": round(len(plain) / 1e3 / code_ms, 1)}), flush=True)


if __name__ == "__main__":
    main()
