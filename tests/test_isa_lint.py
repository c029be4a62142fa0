"""CPU (-m "not gpu"): static checks on the gfx950 code hipcc generates for the specialised kernels (hipcc
cross-compiles without a GPU).  They guard the properties the measured performance depends on and that no
functional test sees: no scratch, no flat_* memory instructions (they would tie vmcnt to lgkmcnt), LDS within
the 160 KiB of a workgroup, registers within the occupancy the shape needs, and the size of the unrolled byte
loop (the path is issue / latency bound: instructions per bit are the figure of merit)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _compile(zlib_, header, waves, tmp_path):
    from zpaq_amd import prebuild
    old = os.environ.get("ZPAQ_AMD_SPEC_WAVES")
    os.environ["ZPAQ_AMD_SPEC_WAVES"] = str(waves)
    try:
        src, _ = prebuild.source_and_key(header)
    finally:
        if old is None:
            del os.environ["ZPAQ_AMD_SPEC_WAVES"]
        else:
            os.environ["ZPAQ_AMD_SPEC_WAVES"] = old
    d = tmp_path / f"w{waves}"
    d.mkdir()
    (d / "k.hip").write_text(src)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-label", "-mllvm",
                    "-simplifycfg-sink-common=false", "-I", os.path.join(ROOT, "zpaq_amd", "csrc", "device"), "--genco",
                    "k.hip", "-o", "k.hsaco", "-save-temps"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    asm = (d / "k-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
    shutil.rmtree(d)
    return asm


def _kernels(asm):
    meta = asm[asm.index("amdhsa.kernels:"):]
    out = {}
    for entry in [e for e in meta.split("\n  - ")[1:] if ".name:" in e]:
        g = lambda k: re.search(rf"\.{k}:\s+(\S+)", entry).group(1)
        out[g("name")] = {k: int(g(k)) for k in ("vgpr_count", "sgpr_count", "group_segment_fixed_size",
                                                  "private_segment_fixed_size", "max_flat_workgroup_size")}
    return out


@pytest.mark.parametrize("waves", [4, 8])
def test_m5_kernel_properties(zlib_, tmp_path, waves):
    from zpaq_amd import corpus
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    header, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    asm = _compile(zlib_, header, waves, tmp_path)
    ks = _kernels(asm)
    assert set(ks) == {"zpq_spec_encode", "zpq_spec_decode"}
    for name, k in ks.items():
        assert k["private_segment_fixed_size"] == 0, (name, "spills to scratch")
        assert k["group_segment_fixed_size"] <= 160 * 1024
        assert k["max_flat_workgroup_size"] == 64 * waves
        assert k["vgpr_count"] <= 512 // (waves // 4), (name, k["vgpr_count"])   # wavefronts per SIMD the shape needs
    code = [l.split()[0] for l in asm.split("\n") if re.match(r"^\t[a-z_0-9]+(\s|$)", l)]
    assert not [op for op in code if op.startswith(("flat_", "scratch_"))]
    # the unrolled byte loop of the encoder, all paths: 4 234 instructions when this test was written
    enc = asm[asm.index("zpq_spec_encode:"):]
    enc = enc[:enc.index("s_endpgm")].split("\n")
    start = next(i for i, l in enumerate(enc) if "Loop Header: Depth=1" in l and i > len(enc) // 4)
    loop = [l for l in enc[start:] if re.match(r"^\t[a-z_0-9]+(\s|$)", l)]
    assert 3000 < len(loop) < 4600, len(loop)


def test_m5_two_blocks_per_wavefront_decoder_properties(zlib_, tmp_path):
    """device/spec_dual_kernel.h: one workgroup of four wavefronts per CU (the LDS is full), so up to 512 registers per
    lane would do -- what matters is no scratch, no flat accesses, and that the byte loop serves TWO blocks with about the
    instructions the one-block kernel needs for one (4 495 against 4 329 when this was written: DESIGN.md section 4.2)."""
    from zpaq_amd import corpus, prebuild
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    header, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    src, _ = prebuild.dual_source_and_key(header)
    d = tmp_path / "dual"
    d.mkdir()
    (d / "k.hip").write_text(src)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-label", "-mllvm",
                    "-simplifycfg-sink-common=false", "-I", os.path.join(ROOT, "zpaq_amd", "csrc", "device"), "--genco",
                    "k.hip", "-o", "k.hsaco", "-save-temps"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = (d / "k-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
    shutil.rmtree(d)
    ks = _kernels(asm)
    assert set(ks) == {"zpq_spec_decode2"}
    k = ks["zpq_spec_decode2"]
    assert k["private_segment_fixed_size"] == 0 and k["group_segment_fixed_size"] <= 160 * 1024
    assert k["max_flat_workgroup_size"] == 256 and k["vgpr_count"] <= 256
    code = [l.split()[0] for l in asm.split("\n") if re.match(r"^\t[a-z_0-9]+(\s|$)", l)]
    assert not [op for op in code if op.startswith(("flat_", "scratch_"))]
    body = asm[asm.index("zpq_spec_decode2:"):]
    body = body[:body.index("s_endpgm")].split("\n")
    start = next(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and i > len(body) // 5)
    loop = [l for l in body[start:] if re.match(r"^\t[a-z_0-9]+(\s|$)", l)]
    assert 3500 < len(loop) < 5200, len(loop)


def test_m5_persistent_launch_properties(zlib_, tmp_path):
    """The persistent launch of the headline's chain (device/pipe_persist.h, throughput shape): ONE workgroup per compute unit
    is what the engine sizes the launch by, so the kernel must stay inside a CU's registers for 8 wavefronts (2 per SIMD: at most
    256 VGPRs) without spilling, inside its LDS, and carry only the small stack the per-unit functions need; the streams between
    its units must leave through write-through stores (`sc1`) and be acquired with a `buffer_inv sc1` -- the XCDs' L2s are not
    coherent with each other inside one launch, and no functional test on the CPU can see a plain store there."""
    from zpaq_amd import corpus, prebuild
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    header, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    src, _ = prebuild.pipe_source_and_key(header, 0)
    lds_define = int(re.search(r"#define ZPQ_PERSIST_LDS_BYTES (\d+)", src).group(1))
    d = tmp_path / "persist"
    d.mkdir()
    (d / "k.hip").write_text(src)
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-label", "-mllvm",
                    "-simplifycfg-sink-common=false", "-I", os.path.join(ROOT, "zpaq_amd", "csrc", "device"), "--genco",
                    "k.hip", "-o", "k.hsaco", "-save-temps"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    asm = (d / "k-hip-amdgcn-amd-amdhsa-gfx950.s").read_text()
    shutil.rmtree(d)
    k = _kernels(asm)["zpq_pipe_persist"]
    assert k["max_flat_workgroup_size"] == 512
    assert k["vgpr_count"] <= 256, k
    assert k["group_segment_fixed_size"] == lds_define <= 160 * 1024, (k, lds_define)
    assert k["private_segment_fixed_size"] <= 256, k              # (call frames of the per-unit functions, no spilled state)
    entry = asm[asm.index("amdhsa.kernels:"):]
    entry = [e for e in entry.split("\n  - ")[1:] if ".name:" in e and "zpq_pipe_persist" in e][0]
    assert re.search(r"\.vgpr_spill_count:\s+0\b", entry) and re.search(r"\.sgpr_spill_count:\s+0\b", entry), entry
    body = asm[:asm.index("amdhsa.kernels:")]
    # the code of the launch: the kernel and the unit functions it calls (label ... .Lfunc_end)
    funcs = re.findall(r"^(\S+):[ \t]*; @\S+\n(.*?)^\.Lfunc_end\d+:", body, re.S | re.M)
    persist = "\n".join(text for name, text in funcs if "pipe_persist" in name)
    assert persist, "no persistent code found in the assembly"
    assert len(re.findall(r"buffer_store_dword\S* .* sc1", persist)) >= 20, "stream stores of the persistent launch must be write-through"
    assert "buffer_inv sc1" in persist
    # flat_* instructions tie vmcnt to lgkmcnt: the unit functions take the launch arguments by reference (a generic pointer
    # to the kernel's copy), so each reads them with a handful of flat loads at its start and once per 512-byte chunk -- fine --
    # but the per-byte loops must not contain any: a unit function with more than a few dozen would have them in a loop body
    for name, text in funcs:
        if "pipe_persist" in name:
            assert len(re.findall(r"\bflat_(?:load|store)", text)) <= 32, name
