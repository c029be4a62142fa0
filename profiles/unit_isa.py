#!/usr/bin/env python3
"""Static instruction counts of the UNIT FUNCTIONS of a chain's persistent encoder launch (no GPU needed).

    python profiles/unit_isa.py [method=3] [kind=lcg] [block_bytes=262144] [mode: 0 throughput | 1 latency]

Compiles the generated source of the chain's pipelined encoder for gfx950 with the engine's flags, cuts the assembly into the
per-unit functions (pipe_persist_unit<Chain, kind, role>: one real function per unit) and reports, for the innermost loop
nest that is the unit's per-BYTE loop, the instructions by class and per coded bit.  A lane-per-block unit issues one
instruction per >= 4 cycles (wave64 on a 16-lane SIMD) and its 8 bits are a dependent chain, so `instructions per bit x ~5`
is the floor of its time per bit when nothing waits for memory -- the number the unit profile (ZPAQ_AMD_PERSIST_PROF) is to be
read against."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def classify(op):
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_call")): return "branch"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"


def main():
    import zpaq_amd as z
    from zpaq_amd import corpus, prebuild
    method = sys.argv[1] if len(sys.argv) > 1 else "3"
    kind = sys.argv[2] if len(sys.argv) > 2 else "lcg"
    bs = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 18
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    if method.startswith("L"):
        header = z.builtin_model_header(int(method[1:]))
    else:
        header = z.method_to_header(z.expand_method(method, corpus.block(kind, bs, corpus.BASE_SEED)))[0]
    src, key = prebuild.pipe_source_and_key(header, mode)
    inc = os.path.join(ROOT, "zpaq_amd", "csrc", "device")
    with tempfile.TemporaryDirectory() as td:
        hip = os.path.join(td, "k.hip")
        open(hip, "w").write(src)
        subprocess.run([prebuild.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-label", "-mllvm",
                        "-simplifycfg-sink-common=false", "-I", inc, "--genco", hip, "-o", os.path.join(td, "k.hsaco"),
                        "-save-temps"], cwd=td, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        asm = open(os.path.join(td, "k-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    names = {0: "HCOMP", 1: "ROW", 2: "light", 3: "ICM map", 4: "ISSE map", 5: "MIX"}
    lk = re.search(r"LIGHT_KIND\[\d+\] = \{([^}]*)\}", src)
    light_kinds = [int(x) for x in lk.group(1).split(",")] if lk else []
    lname = {2: "CONS", 3: "CM", 4: "MATCH", 5: "AVG", 6: "MIX2", 7: "SSE", 8: "CODER", 9: "CM bits", 10: "MIX2 bits", 11: "SSE bits"}
    print(f"chain of method {method} ({kind}, {bs} B blocks), n = {header[6]} components, shape {mode}; key {key}")
    print(f"{'unit':28s} {'loop':>6s} {'per bit':>8s}   valu  salu  vmem   lds  wait  branch  nop   (instructions of the per-byte loop nest, static, all paths)")
    # functions: from a symbol line "<name>:" to its ".Lfunc_end"
    for m in re.finditer(r"^(_ZN3zpq17pipe_persist_unitIN7zpq_gen6ChainPELi(\d+)ELi(\d+)EEEvRKNS_8PipeArgsEjii):[^\n]*\n(.*?)^\.Lfunc_end", asm, re.S | re.M):
        kindn, role, body = int(m.group(2)), int(m.group(3)), m.group(4)
        lines = body.split("\n")
        # the per-byte loop: the loop (any depth) with the most instructions that is not the chunk loop itself -- take the
        # largest Depth >= 2 loop header region up to the next header of the same or lower depth
        heads = [(i, int(re.search(r"Depth=(\d+)", l).group(1))) for i, l in enumerate(lines) if "Loop Header: Depth=" in l]
        best = None
        for hi, (i, d) in enumerate(heads):
            if d < 2:
                continue
            j = next((i2 for i2, d2 in heads[hi + 1:] if d2 <= d), len(lines))
            ops = [l.split()[0] for l in lines[i:j] if re.match(r"^\s+[a-z_0-9]+(\s|$)", l) and not l.strip().startswith((";", "."))]
            if best is None or len(ops) > len(best[0]):
                best = (ops, d)
        if best is None:
            continue
        ops = best[0]
        c = collections.Counter(classify(o) for o in ops)
        tag = names.get(kindn, "?")
        if kindn == 2 and role < len(light_kinds):
            tag = lname.get(light_kinds[role], "light")
        print(f"{tag + ' (role %d)' % role:28s} {len(ops):6d} {len(ops) / 8:8.1f}   {c['valu']:4d}  {c['salu']:4d}  {c['vmem']:4d}  {c['lds']:4d}  {c['s_waitcnt']:4d}  {c['branch']:6d}  {c['s_nop']:3d}")


if __name__ == "__main__":
    main()
