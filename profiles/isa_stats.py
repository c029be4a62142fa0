#!/usr/bin/env python3
"""Static instruction statistics of a specialised kernel (no GPU needed).

    python profiles/isa_stats.py [method] [waves]      # default: "5" on 1 MiB text, 4 blocks per workgroup

Asks the library for the generated kernel source of the block header that `method` produces, compiles it for
gfx950 with the flags the engine uses, and reports registers / LDS / scratch and an opcode histogram of the
unrolled byte loop of zpq_spec_encode (everything after the one-off prologue), per coded bit.  This is the
"instructions per bit" figure DESIGN.md section 5 quotes: the path is issue bound, so it is the number to drive down.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import zpaq_amd as z
    from zpaq_amd import corpus, prebuild
    method = sys.argv[1] if len(sys.argv) > 1 else "5"
    waves = sys.argv[2] if len(sys.argv) > 2 else "4"
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    xm = method if method.startswith("x") else z.expand_method(method, blk)
    header, _, _ = z.method_to_header(xm)
    os.environ["ZPAQ_AMD_SPEC_WAVES"] = waves
    src, key = prebuild.source_and_key(header)
    inc = os.path.join(ROOT, "zpaq_amd", "csrc", "device")
    with tempfile.TemporaryDirectory() as td:
        hip = os.path.join(td, "k.hip")
        open(hip, "w").write(src)
        subprocess.run([prebuild.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-label", "-mllvm",
                        "-simplifycfg-sink-common=false", "-I", inc, "--genco", hip, "-o", os.path.join(td, "k.hsaco"),
                        "-save-temps"], cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(td, "k-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    print(f"method {xm}\nkey {key}  blocks/workgroup {waves}")
    meta = asm[asm.index("amdhsa.kernels:"):]
    for entry in [e for e in meta.split("\n  - ")[1:] if ".name:" in e]:
        grab = lambda k: (re.search(rf"\.{k}:\s+(\S+)", entry) or [None, "?"])[1]
        if grab("name").startswith("zpq_spec"):
            print(f"{grab('name')}: vgpr {grab('vgpr_count')} sgpr {grab('sgpr_count')} "
                  f"lds {grab('group_segment_fixed_size')} scratch {grab('private_segment_fixed_size')}")
    body = asm[asm.index("zpq_spec_encode:"):]
    body = body[:body.index("s_endpgm")]
    ins = [l.split()[0] for l in body.split("\n") if re.match(r"^\s+[a-z_0-9]+(\s|$)", l) and not l.strip().startswith(";")]
    # the byte loop starts at the first HCOMP-sized block after the prologue: find it as the last third of the text
    # is not robust; use the loop header label LLVM prints ("Loop Header: Depth=1" of the byte loop)
    lines = body.split("\n")
    start = next(i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l and i > len(lines) // 4)
    loop = [l.split()[0] for l in lines[start:] if re.match(r"^\s+[a-z_0-9]+(\s|$)", l) and not l.strip().startswith(";")]
    print(f"zpq_spec_encode: {len(ins)} instructions, byte loop {len(loop)} (static, all paths) = {len(loop) / 8:.0f} per bit")
    cls = collections.Counter()
    for op in loop:
        if op.startswith("s_nop"):
            cls["hazard s_nop"] += 1
        elif op.startswith("s_waitcnt"):
            cls["s_waitcnt"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cls["global memory"] += 1
        elif op.startswith("ds_"):
            cls["LDS"] += 1
        elif op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.endswith("_dpp") or "dpp" in op:
            cls["cross-lane (readlane/DPP)"] += 1
        elif op.startswith("v_"):
            cls["VALU"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"):
            cls["branch"] += 1
        else:
            cls["SALU"] += 1
    for k, v in cls.most_common():
        print(f"  {k:28s} {v:5d} per byte  {v / 8:6.1f} per bit")
    top = collections.Counter(re.sub(r"_(e32|e64|sdwa|dpp)$", "", op) for op in loop).most_common(14)
    print("  top opcodes per byte: " + ", ".join(f"{k} {v}" for k, v in top))


if __name__ == "__main__":
    main()
