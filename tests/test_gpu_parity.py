"""GPU (-m gpu): the HIP path, called through the C ABI, against the oracle, the
golden vectors made by the reference, and (when oracle/_ref travelled here) the
reference itself.  Bit-exact everywhere: this is integer/byte work."""
import hashlib
import os
import sys

import numpy as np
import pytest

from conftest import b64, gen_input
from oracle.oracle_py import have_ref, parse_block
from zpaq_amd import corpus

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

# 4 = pipelined encoder (decodes with 3), 3 = per-header specialised wavefront kernel, 2 = generic wave-parallel, 1 = generic one-lane
KERNELS = [4, 3, 2, 1]
SIZE_LIMIT = {4: 262144, 3: 262144, 2: 20000, 1: 2000}   # the generic kernels are fallbacks: keep their cases short


def test_cross_lane_selftest(gpu):
    assert gpu.selftest()[:6] == [2016, 21344, 123, 2016, 133, 13671]


def test_native_library_is_the_one_running(gpu):
    maps = open("/proc/self/maps").read()
    assert "libzpaq_amd.so" in maps and gpu.device_count() >= 1


@pytest.mark.parametrize("kernel", KERNELS)
def test_encode_matches_oracle_and_golden(gpu, oracle, golden, kernel):
    """zpq_encode_batch over every small golden case: coded stream == oracle == reference archive slice."""
    gpu.set_kernel(kernel)
    try:
        entries = [e for e in golden["method_cases"]
                   if e["n"] <= min(65536, SIZE_LIMIT[kernel]) and bytes.fromhex(e["header"])[6]]
        plans, inputs = [], []
        cache = {}
        for e in entries:
            hdr = bytes.fromhex(e["header"])
            if hdr not in cache:
                cache[hdr] = gpu.Plan(hdr)
            plans.append(cache[hdr])
            inputs.append(b"\0" + gen_input(e).tobytes())
        coded = gpu.encode_batch(plans, inputs)
        bad = []
        for e, inp, c in zip(entries, inputs, coded):
            hdr = bytes.fromhex(e["header"])
            if c != oracle.encode(hdr, inp):
                bad.append((e["kind"], e["n"], e["method"], hdr[6]))
            elif "archive_b64" in e:
                a = b64(e)
                ps = e["payload_start"]
                if a[ps:ps + len(c) + 4] != c + b"\0\0\0\0":
                    bad.append(("archive", e["kind"], e["n"], e["method"]))
        assert not bad, bad
    finally:
        gpu.set_kernel(0)


@pytest.mark.parametrize("kernel", KERNELS)
def test_compress_blocks_bit_identical_archives(gpu, golden, kernel):
    """Batched compressBlock: whole archives (tag .. 255) hash-identical to the reference's."""
    gpu.set_kernel(kernel)
    try:
        lim = SIZE_LIMIT[kernel]
        by_method = {}
        for e in golden["method_cases"]:
            if e["n"] <= lim:
                by_method.setdefault(e["method"], []).append(e)
        for method, es in by_method.items():
            archives = gpu.compress_blocks([gen_input(e) for e in es], method, [e["filename"] for e in es],
                                           [e["comment"] for e in es])
            for e, a in zip(es, archives):
                assert len(a) == e["len"] and hashlib.sha1(a).hexdigest() == e["sha1"], (e["kind"], e["n"], method)
    finally:
        gpu.set_kernel(0)


@pytest.mark.parametrize("kernel", KERNELS)
def test_all_nine_component_types(gpu, oracle, golden, kernel):
    gpu.set_kernel(kernel)
    try:
        e = golden["config_cases"][0]
        a = b64(e)
        hdr = bytes.fromhex(e["header"])
        d = gen_input(e).tobytes()
        plan = gpu.Plan(hdr)
        c = gpu.encode_batch([plan], [b"\0" + d])[0]
        ps = e["payload_start"]
        assert a[ps:ps + len(c) + 4] == c + b"\0\0\0\0"
        (dec, used), = gpu.decode_batch([plan], [a[ps:]], [len(d) + 64])
        assert dec == b"\0" + d and used == len(c) + 4
    finally:
        gpu.set_kernel(0)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("idx", [0, 1, 2])
def test_legacy_min_mid_max_models(gpu, golden, kernel, idx):
    """BASELINE.json names mid.cfg / max.cfg: the built-in chains, encode + decode, from reference archives."""
    if kernel == 1 and idx != 1:
        pytest.skip("one-lane fallback kernel: one legacy model is enough")
    gpu.set_kernel(kernel)
    try:
        e = golden["level_cases"][idx]
        a = b64(e)
        hdr = bytes.fromhex(e["header"])
        d = gen_input(e).tobytes()
        plan = gpu.Plan(hdr)
        ps = e["payload_start"]
        c = gpu.encode_batch([plan], [b"\0" + d])[0]
        assert a[ps:ps + len(c) + 4] == c + b"\0\0\0\0"
        assert gpu.decompress(a) == d
    finally:
        gpu.set_kernel(0)


@pytest.mark.parametrize("level,kind,nblocks,bs", [(2, "lcg", 64, 1 << 18), (3, "text", 8, 1 << 20)])
def test_legacy_models_at_baseline_block_sizes(gpu, level, kind, nblocks, bs):
    """mid.cfg on configs[1]'s 256 KiB LCG blocks, max.cfg on configs[2]'s 1 MiB text blocks
    (Compressor::startBlock(2 | 3), libzpaq.cpp:2793-2839) -- coded through the PERSISTENT launch of the pipelined encoder,
    every coded payload against what the reference made of the same block (tests/golden/legacy_sha1.json, written by
    tests/golden/make_legacy_golden.py from oracle/_ref), decoded back by the lockstep decoder."""
    import json
    gj = json.load(open(os.path.join(ROOT, "tests", "golden", "legacy_sha1.json")))["levels"][str(level)]
    assert gj["block_bytes"] == bs and f"'{kind}'" in gj["corpus"]
    blocks = [corpus.block(kind, bs, corpus.BASE_SEED + b).tobytes() for b in range(nblocks)]
    hdr = gpu.builtin_model_header(level)
    plan = gpu.Plan(hdr)
    coded = gpu.encode_batch([plan] * nblocks, [b"\0" + d for d in blocks])
    assert gpu.lib().zpq_last_persistent() == 1, "the built-in model's chain did not take the persistent launch"
    for b, c in enumerate(coded):
        assert len(c) == gj["blocks"][b]["coded_len"], (level, b)
        assert hashlib.sha1(c + b"\0\0\0\0").hexdigest() == gj["blocks"][b]["payload_sha1"], (level, b)
    gpu.set_kernel(6)                      # the lockstep decoder, whatever the batch size
    cap = 262144 + 9                       # (every block's first 256 KiB: a decode's time is the bytes it decodes)
    try:
        back = gpu.decode_batch([plan] * nblocks, [c + b"\0\0\0\0" for c in coded], [min(len(d) + 9, cap) for d in blocks])
    finally:
        gpu.set_kernel(0)
    for b, (d, consumed) in enumerate(back):
        assert d == (b"\0" + blocks[b])[:cap], (level, b)


@pytest.mark.parametrize("kernel", KERNELS)
def test_decode_reference_archives(gpu, golden, kernel):
    gpu.set_kernel(kernel)
    try:
        es = [e for e in golden["method_cases"] if "archive_b64" in e]
        stream = b"".join(b64(e) for e in es)           # multi-block archive, decoded as one batch
        want = b"".join(gen_input(e).tobytes() for e in es)
        assert gpu.decompress(stream) == want
    finally:
        gpu.set_kernel(0)


def test_specialised_kernel_is_the_one_running(gpu, golden, oracle):
    """Standard chains come from the in-tree code-object cache; data-dependent ones through hipRTC."""
    import ctypes as C
    L = gpu.lib()
    note = C.create_string_buffer(4096)
    d = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    h, _, _ = gpu.method_to_header(gpu.expand_method("5", d))
    plan = gpu.Plan(h)
    # compression runs on the pipelined encoder (4), decompression on the per-header wavefront kernel (3)
    assert L.zpq_plan_kernel_kind(plan._h, note, 4096) == 4, note.value
    assert note.value.startswith(b"cache:"), note.value
    assert L.zpq_plan_kernel_kind2(plan._h, 1, note, 4096) == 3, note.value
    assert note.value.startswith(b"cache:"), note.value
    d = corpus.block("records", 30000, 77)          # period detection -> a chain nobody prebuilt
    h2, _, _ = gpu.method_to_header(gpu.expand_method("5", d))
    assert h2 != h
    plan2 = gpu.Plan(h2)
    for dec, want in ((0, 4), (1, 3)):
        kind = L.zpq_plan_kernel_kind2(plan2._h, dec, note, 4096)
        assert kind == want, note.value
        assert note.value.startswith(b"hiprtc") or note.value.startswith(b"cache:")
    # a header that certainly was not prebuilt (no build step knows this chain): must come out of hipRTC, and
    # code correctly in both directions
    h3, _, _ = gpu.method_to_header("x0,0ci2,1,1c0,3m16s")
    plan3 = gpu.Plan(h3)
    assert L.zpq_plan_kernel_kind(plan3._h, note, 4096) == 4, note.value
    assert note.value.startswith(b"hiprtc"), note.value
    assert L.zpq_plan_kernel_kind2(plan3._h, 1, note, 4096) == 3, note.value
    assert note.value.startswith(b"hiprtc"), note.value
    d3 = b"\0" + corpus.block("text", 50000, 4242).tobytes()
    c3 = gpu.encode_batch([plan3], [d3])[0]
    assert c3 == oracle.encode(h3, d3)
    (back, used), = gpu.decode_batch([plan3], [c3 + b"\0\0\0\0"], [len(d3) + 8])
    assert back == d3 and used == len(c3) + 4


def test_decoder_status_codes(gpu, golden, oracle):
    e = [x for x in golden["method_cases"] if x["kind"] == "text" and x["n"] == 20000 and x["method"] == "5"][0]
    hdr = bytes.fromhex(e["header"])
    plan = gpu.Plan(hdr)
    d = gen_input(e).tobytes()
    c = gpu.encode_batch([plan], [b"\0" + d])[0]
    # truncated input -> EOF; "decode first k bytes" -> OK with consumed == 0
    res, st = gpu.decode_batch([plan], [c[:len(c) // 2]], [len(d) + 8], check=False)
    assert st[0] in (6, 2)
    (dec, used), = gpu.decode_batch([plan], [c + b"\0\0\0\0"], [1001])
    assert used == 0 and dec == (b"\0" + d)[:1001]
    # damaged streams: Decoder::decode's contract is "archive corrupted" as soon as curr leaves [low, high]
    # (libzpaq.cpp:2108) or the end-of-stream flag is followed by non-zero bytes (2134), "unexpected end of file"
    # when the input runs out (2120); otherwise it decodes garbage without complaint.  Whatever the reference
    # algorithm does with a given damage, the device decoder must do exactly the same: same status, same bytes.
    good = c + b"\0\0\0\0"
    damaged, outcomes = [], []
    for pos, bit in [(0, 7), (3, 0), (7, 6), (100, 3), (len(c) // 2, 1), (len(c) - 9, 5), (len(c) - 2, 2), (len(c), 0)]:
        bad = bytearray(good)
        bad[pos] ^= 1 << bit
        damaged.append(bytes(bad))
        outcomes.append(oracle.decode_outcome(hdr, bytes(bad), len(d) + 64))
    damaged.append(good[:-6]); outcomes.append(oracle.decode_outcome(hdr, good[:-6], len(d) + 64))
    res, st = gpu.decode_batch([plan] * len(damaged), damaged, [len(d) + 64] * len(damaged), check=False)
    for (dec, _), got, (want_st, want) in zip(res, st, outcomes):
        assert got == want_st, (got, want_st)
        if want_st == 0:
            assert dec == want
    assert {o[0] for o in outcomes} >= {2}, outcomes      # the sample does contain detected corruption
    # output overflow on encode is reported, not silently truncated
    outs, st, lens = gpu.encode_batch([plan], [b"\0" + d], out_cap=[100], check=False)
    assert st[0] == 3 and lens[0] == len(c)


def test_mixed_plans_in_one_batch_and_ragged_sizes(gpu, oracle):
    """Blocks with different chains (period detection!) and ragged sizes, incl. empty, in ONE batch."""
    blocks = [corpus.block("records", 30000, 5), corpus.block("text", 0, 6), corpus.block("text", 1, 7),
              corpus.block("lcg", 12345, 8), corpus.block("pattern", 4097, 9), corpus.block("zeros", 70000, 10)]
    archives = gpu.compress_blocks(blocks, "5")
    ncomps = set()
    for d, a in zip(blocks, archives):
        f = parse_block(a)
        ncomps.add(f["header"][6])
        coded = oracle.encode(f["header"], b"\0" + d.tobytes())
        ps = f["payload_start"]
        assert a[ps:ps + len(coded) + 4] == coded + b"\0\0\0\0"
    assert len(ncomps) >= 2
    assert gpu.decompress(b"".join(archives)) == b"".join(b.tobytes() for b in blocks)


def test_full_size_blocks_roundtrip_and_reference_cross_check(gpu):
    """BASELINE block size (1 MiB, -m5): known-answer SHA-1 from BASELINE.md §2, round trip, and the
    reference decoding our archive / producing the same bytes when oracle/_ref is here."""
    blocks = [corpus.block("lcg", 1 << 20, corpus.BASE_SEED), corpus.block("text", 1 << 20, corpus.BASE_SEED)]
    archives = gpu.compress_blocks(blocks, "5")
    assert hashlib.sha1(archives[0]).hexdigest() == "6e0850c1a89c9647a9ba954eaf143d297c8cf2d5" and len(archives[0]) == 1049128
    back = gpu.decompress(b"".join(archives))
    assert back == b"".join(b.tobytes() for b in blocks)
    if have_ref():
        from oracle.oracle_py import Ref
        ref = Ref()
        assert ref.compress_block(blocks[1], "5") == archives[1]
        assert ref.decompress(archives[1], 1 << 20) == blocks[1].tobytes()


def test_zeros_known_answers(gpu):
    """Highly compressible blocks at 64 KiB and 1 MiB: BASELINE.md §2 SHA-1s (generator independent)."""
    a64, a1m = gpu.compress_blocks([np.zeros(65536, np.uint8), np.zeros(1 << 20, np.uint8)], "5")
    assert len(a64) == 317 and hashlib.sha1(a64).hexdigest() == "071deeac62e6fc28932fe84d52c26b7b6debe788"
    assert len(a1m) == 341 and hashlib.sha1(a1m).hexdigest() == "33f41ec44b376e492954d759ddd560f07ee7734b"


def test_decode_reference_archives_with_preprocessing(gpu):
    """Reference archives of methods 3 and 4 (LZ77 / BWT / E8E9 pre-processing + a context model):
    the model is decoded on the GPU, the PCOMP program from the archive is run on the host."""
    if not have_ref():
        pytest.skip("oracle/_ref not present")
    from oracle.oracle_py import Ref
    ref = Ref()
    parts, stream = [], b""
    for kind, n, m in [("lcg", 40000, "3"), ("text", 60000, "3"), ("text", 50000, "3,128,1"), ("records", 30000, "4,30,0"),
                       ("text", 30000, "4,128,3"), ("pattern", 20000, "3,200,2")]:
        d = corpus.block(kind, n, 99)
        parts.append(d.tobytes())
        stream += ref.compress_block(d, m)
    assert gpu.decompress(stream) == b"".join(parts)


def test_random_hcomp_programs_all_kernels(gpu, oracle, golden):
    """HCOMP -> HIP translation (hipRTC path) and both interpreters against the reference's archives."""
    for kernel in KERNELS:
        gpu.set_kernel(kernel)
        try:
            cases = golden["vm_cases"] if kernel != 3 else golden["vm_cases"][:6]    # ~3 s of hipRTC each
            plans = [gpu.Plan(bytes.fromhex(e["header"])) for e in cases]
            inputs = [b"\0" + gen_input(e).tobytes() for e in cases]
            coded = gpu.encode_batch(plans, inputs)
            bad = []
            for e, c in zip(cases, coded):
                a = b64(e)
                ps = e["payload_start"]
                if a[ps:ps + len(c) + 4] != c + b"\0\0\0\0":
                    bad.append(e["name"])
            assert not bad, (kernel, bad)
            res = gpu.decode_batch(plans, [b64(e)[e["payload_start"]:] for e in cases], [e["n"] + 64 for e in cases])
            for e, (dec, used) in zip(cases, res):
                assert dec == b"\0" + gen_input(e).tobytes(), (kernel, e["name"])
        finally:
            gpu.set_kernel(0)


def test_more_than_64_components_uses_the_generic_kernel(gpu, golden):
    import ctypes as C
    e = [c for c in golden["config_cases"] if c["name"] == "seventy_components"][0]
    plan = gpu.Plan(bytes.fromhex(e["header"]))
    note = C.create_string_buffer(512)
    assert gpu.lib().zpq_plan_kernel_kind(plan._h, note, 512) == 1
    a = b64(e)
    ps = e["payload_start"]
    d = gen_input(e).tobytes()
    c = gpu.encode_batch([plan], [b"\0" + d])[0]
    assert a[ps:ps + len(c) + 4] == c + b"\0\0\0\0"
    assert gpu.decompress(a) == d


DEEP_ISSE = "x0,0ci1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1m"   # 1 ICM + 17 ISSE: 35 KiB of side tables, more than a block's LDS


@pytest.mark.parametrize("waves", ["4", "8"])
def test_both_workgroup_shapes_and_arena_side_tables(gpu, oracle, golden, monkeypatch, waves):
    """The specialised kernel exists in two workgroup shapes (4 / 8 blocks per workgroup; the engine picks by batch
    size).  Force each one and check it bit-exact: standard chains (in the 8-block shape half of the -m5 side
    tables live in the arena and are fetched a bit ahead), and a chain too deep for LDS in either shape."""
    monkeypatch.setenv("ZPAQ_AMD_SPEC_WAVES", waves)
    gpu.set_kernel(3)
    try:
        entries = [e for e in golden["method_cases"] if e["n"] <= 65536 and bytes.fromhex(e["header"])[6]]
        entries += [golden["config_cases"][0], golden["level_cases"][2]]
        plans, inputs, cache = [], [], {}
        for e in entries:
            hdr = bytes.fromhex(e["header"])
            if hdr not in cache:
                cache[hdr] = gpu.Plan(hdr)
            plans.append(cache[hdr])
            inputs.append(b"\0" + gen_input(e).tobytes())
        deep_hdr, _, _ = gpu.method_to_header(DEEP_ISSE)
        deep = gpu.Plan(deep_hdr)
        for k, kind in enumerate(["text", "records", "lcg", "text"]):
            plans.append(deep)
            inputs.append(b"\0" + corpus.block(kind, 30000 + 1111 * k, 900 + k).tobytes())
        hdrs = [bytes.fromhex(e["header"]) for e in entries] + [deep_hdr] * 4
        coded = gpu.encode_batch(plans, inputs)
        bad = [i for i, (h, inp, c) in enumerate(zip(hdrs, inputs, coded)) if c != oracle.encode(h, inp)]
        assert not bad, bad
        back = gpu.decode_batch(plans, [c + b"\0\0\0\0" for c in coded], [len(x) + 64 for x in inputs])
        assert all(dec == inp for (dec, _), inp in zip(back, inputs))
    finally:
        gpu.set_kernel(0)


def test_torch_corpus_matches_numpy(gpu):
    """bench.py generates its text corpus on the GPU; it must be the same bytes as the numpy generator.
    Runs in its own process, torch first, the way bench.py orders things (torch brings its own HIP runtime and
    must initialise before libzpaq_amd.so is loaded)."""
    import subprocess
    code = (
        "import torch, sys; sys.path.insert(0, %r)\n"
        "from zpaq_amd import corpus, corpus_torch\n"
        "o = corpus_torch.text_blocks(3, 200000, corpus.BASE_SEED + 40, torch.device('cuda', 0), chunk=2).cpu().numpy()\n"
        "assert all((o[b] == corpus.zipf_text(200000, corpus.BASE_SEED + 40 + b)).all() for b in range(3))\n"
        "print('same')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "same" in r.stdout, r.stderr[-2000:]


def test_device_resident_entry_point_timed_and_in_flight(gpu):
    """zpq_code_device_multi on buffers already in HBM (what bench.py times), in both forms: timed = 1 waits for the results and
    takes the persistent launch; timed = 0 returns with the work in flight on the caller's stream and runs the step kernels --
    in the shape chosen FOR the step kernels (engine.cpp pipe_mode_for: blocks of 128 KiB and more in a small batch take the
    2048-byte steps).  Same coded bytes either way, and the same as zpq_encode_batch's.  Own process: torch (device buffers)
    has to initialise HIP before the library is loaded."""
    import subprocess
    code = (
        "import torch, sys, ctypes as C; sys.path.insert(0, %r)\n"
        "import numpy as np, zpaq_amd as z\nfrom zpaq_amd import corpus\n"
        "dev = torch.device('cuda', 0)\nz.init(0)\nL = z.lib()\n"
        "blocks = [corpus.block(['text', 'records'][i %% 2], 160000 + 37 * i, 800 + i) for i in range(48)]\n"
        "hdrs = [z.method_to_header(z.expand_method('5', b))[0] for b in blocks]\n"
        "plans = {}\n"
        "for h in hdrs: plans.setdefault(h, z.Plan(h))\n"
        "pl = [plans[h] for h in hdrs]\n"
        "ins = [b'\\0' + b.tobytes() for b in blocks]\n"
        "want = z.encode_batch(pl, ins)\n"
        "n, si, so = len(ins), 163840, 208896\n"
        "host = np.zeros((n, si), np.uint8)\n"
        "for i, x in enumerate(ins): host[i, :len(x)] = np.frombuffer(x, np.uint8)\n"
        "d_in = torch.from_numpy(host).to(dev)\n"
        "L.zpq_code_device_multi.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p, C.c_int]\n"
        "PA = (C.c_void_p * n)(*[p._h for p in pl]); IO = (C.c_uint64 * n)(*[i * si for i in range(n)]); IL = (C.c_uint32 * n)(*[len(x) for x in ins])\n"
        "OO = (C.c_uint64 * n)(*[i * so for i in range(n)]); OC = (C.c_uint32 * n)(*[so] * n)\n"
        "for timed in (1, 0):\n"
        "    d_out = torch.zeros((n, so), dtype=torch.uint8, device=dev); d_res = torch.zeros((n, 4), dtype=torch.int32, device=dev)\n"
        "    rc = L.zpq_code_device_multi(0, PA, C.c_void_p(d_in.data_ptr()), IO, IL, n, C.c_void_p(d_out.data_ptr()), OO, OC, C.c_void_p(d_res.data_ptr()), None, timed)\n"
        "    assert rc == 0, L.zpq_last_error()\n"
        "    torch.cuda.synchronize()\n"
        "    res = d_res.cpu().numpy(); out = d_out.cpu().numpy()\n"
        "    assert (res[:, 2] == 0).all()\n"
        "    assert all(out[i, :res[i, 0]].tobytes() == want[i] for i in range(n)), timed\n"
        "    print('FORM', timed, 'persistent', L.zpq_last_persistent() if timed else '-')\n"
        "print('same')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "same" in r.stdout and "FORM 1 persistent 1" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


def test_large_batch_picks_its_own_kernels(gpu, oracle):
    """More than 4 x CUs blocks in ONE call, nothing forced: compression runs on the pipelined encoder (35 groups, the
    last one ragged), decompression picks the 8-blocks-per-workgroup shape of the wavefront kernel by itself
    (engine: dense = blocks > 4 x CUs).  All 1100 coded streams must also equal what the wavefront ENCODER makes
    (two independent kernels), and a sample must equal the oracle."""
    kinds = ["text", "lcg", "records", "zeros"]
    blocks = [corpus.block(kinds[i % 4], 16384 - (i % 7) * 5, 7000 + i) for i in range(1100)]
    arch = gpu.compress_blocks(blocks, "5")
    gpu.set_kernel(3)
    try:
        arch3 = gpu.compress_blocks(blocks, "5")
    finally:
        gpu.set_kernel(0)
    assert arch == arch3
    for i in list(range(0, 1100, 61)) + [1023, 1024, 1087, 1088, 1099]:
        f = parse_block(arch[i])
        coded = oracle.encode(f["header"], b"\0" + blocks[i].tobytes())
        ps = f["payload_start"]
        assert arch[i][ps:ps + len(coded) + 4] == coded + b"\0\0\0\0", i
    assert gpu.decompress(b"".join(arch)) == b"".join(b.tobytes() for b in blocks)


def test_two_blocks_per_wavefront_decoder(gpu, golden):
    """device/spec_dual_kernel.h, which a dense decode launch takes by itself: forced here (zpq_set_kernel(5)) on every
    golden archive whose chain it accepts (the reference wrote them; zpq_decompress checks their SHA-1 trailers) and on a
    batch that has an odd number of blocks, an empty one and a one-byte one -- the same bytes as the one-block kernel (3)
    and the engine's own choice return."""
    tried = 0
    gpu.set_kernel(5)
    try:
        for sect in ("config_cases", "level_cases", "vm_cases", "method_cases"):
            for e in golden[sect]:
                hdr = bytes.fromhex(e["header"])
                if "archive_b64" not in e or hdr[6] == 0 or hdr[6] > 32:
                    continue
                want = gen_input(e).tobytes()
                assert gpu.decompress(b64(e), cap=len(want) + 64) == want, (sect, e.get("name") or e.get("method"))
                tried += 1
    finally:
        gpu.set_kernel(0)
    assert tried >= 40
    blocks = [corpus.block("text", 20000 + 77 * i, 300 + i).tobytes() for i in range(9)] + [b"", b"x"]
    arch = b"".join(gpu.compress_blocks(blocks, "5"))
    outs = []
    for kernel in (5, 3, 0):
        gpu.set_kernel(kernel)
        try:
            outs.append(gpu.decompress(arch))
        finally:
            gpu.set_kernel(0)
    assert outs[0] == outs[1] == outs[2] == b"".join(blocks)


def test_lockstep_decoder(gpu, golden):
    """device/spec_team_kernel.h (row + mixer wavefronts, the 8 blocks of a workgroup bit by bit together), which a dense
    decode launch takes by itself: forced here (zpq_set_kernel(6)) on every golden archive whose chain it accepts (the
    reference wrote them; zpq_decompress checks their SHA-1 trailers), on a batch with an odd number of blocks, an empty
    and a one-byte one, and on 2 048 blocks in one launch (the configs[4] operating point: every CU holds a workgroup) --
    the same bytes as the one-block kernel (3), the two-block kernel (5) and the engine's own choice return."""
    tried = 0
    gpu.set_kernel(6)
    try:
        for sect in ("config_cases", "level_cases", "vm_cases", "method_cases"):
            for e in golden[sect]:
                hdr = bytes.fromhex(e["header"])
                if "archive_b64" not in e or hdr[6] == 0 or hdr[6] > 32:
                    continue
                want = gen_input(e).tobytes()
                assert gpu.decompress(b64(e), cap=len(want) + 64) == want, (sect, e.get("name") or e.get("method"))
                tried += 1
    finally:
        gpu.set_kernel(0)
    assert tried >= 40
    blocks = [corpus.block("text", 20000 + 77 * i, 300 + i).tobytes() for i in range(9)] + [b"", b"x"]
    arch = b"".join(gpu.compress_blocks(blocks, "5"))
    outs = []
    for kernel in (6, 5, 3, 0):
        gpu.set_kernel(kernel)
        try:
            outs.append(gpu.decompress(arch))
        finally:
            gpu.set_kernel(0)
    assert outs[0] == outs[1] == outs[2] == outs[3] == b"".join(blocks)
    many = [corpus.block(("text", "records", "lcg", "zeros")[i % 4], 3000 + (i * 37) % 1500, 900 + i).tobytes() for i in range(2048)]
    arch = b"".join(gpu.compress_blocks(many, "5"))
    for kernel in (6, 5):
        gpu.set_kernel(kernel)
        try:
            assert gpu.decompress(arch) == b"".join(many), kernel
        finally:
            gpu.set_kernel(0)


def test_lockstep_decoder_on_the_mixed_corpus_at_its_operating_point(gpu):
    """BASELINE configs[4] at a quarter of its block size: 2 048 blocks of 256 KiB of configs[3]'s mixed corpus (text, text, LCG,
    records: two chains, n = 23 and the records blocks' chain with their detected periods) coded by the persistent encoder in
    two rounds, decoded in ONE launch that fills the GPU -- the engine's own choice of decoders, then the lockstep decoder
    forced -- every byte compared."""
    mix = ("text", "text", "lcg", "records")
    n, bs = 2048, 256 << 10
    blocks = [corpus.block(mix[i % 4], bs, corpus.BASE_SEED + i) for i in range(n)]
    want = hashlib.sha1(b"".join(b.tobytes() for b in blocks)).hexdigest()
    arch = b"".join(gpu.compress_blocks(blocks, "5"))
    for kernel in (0, 6):
        gpu.set_kernel(kernel)
        try:
            assert hashlib.sha1(gpu.decompress(arch)).hexdigest() == want, kernel
        finally:
            gpu.set_kernel(0)


def test_4_mib_zeros_known_answer(gpu):
    """4 MiB zeros, method 5 -> 410 B (175 MiB of model state per block)."""
    a, = gpu.compress_blocks([np.zeros(4 << 20, np.uint8)], "5")
    assert len(a) == 410 and hashlib.sha1(a).hexdigest() == "27a7b8ea100078baeb624dec8aeffb3bef874e75"
    # (the first 512 KiB decode back: one wavefront needs a minute for the whole block, and the suite has a clock)
    f = parse_block(a)
    (dec, used), = gpu.decode_batch([gpu.Plan(f["header"])], [a[f["payload_start"]:]], [(1 << 19) + 1])
    assert dec == bytes((1 << 19) + 1)


def test_16_mib_zeros_known_answer(gpu):
    """BASELINE.md section 2, the top of the north-star range: 16 MiB zeros, method 5 -> 688 B (463 MiB of model state for
    the one block; one lane per unit of the encoder in latency mode, 32 784 steps).  The first 64 KiB decode back
    (Decompresser::decompress(n): a whole-block decode of one wavefront would take minutes)."""
    a, = gpu.compress_blocks([np.zeros(16 << 20, np.uint8)], "5")
    assert len(a) == 688 and hashlib.sha1(a).hexdigest() == "aecc5f154175bc56a6af2d0015f1e01acef55655"
    f = parse_block(a)
    (dec, used), = gpu.decode_batch([gpu.Plan(f["header"])], [a[f["payload_start"]:]], [65537])
    assert dec == bytes(65537) and used == 0          # PP byte + 64 KiB of zeros, stopped before the end of the stream


def test_one_mib_records_block_with_detected_periods(gpu, ref):
    """512 KiB (1 MiB until round 5: the suite has a clock) of 16-byte records: level-5 period detection adds components (n = 31), a chain no build step knows
    (hipRTC for both the pipelined encoder and the wavefront decoder).  Archive must equal the reference's."""
    d = corpus.block("records", 1 << 19, corpus.BASE_SEED + 3)
    a, = gpu.compress_blocks([d], "5")
    assert parse_block(a)["header"][6] > 23
    assert a == ref.compress_block(d, "5")
    assert gpu.decompress(a) == d.tobytes()


_REF_CACHE = {}


def test_mixed_corpus_batch_against_the_reference(gpu, ref):
    """BASELINE configs[3] in small: block b is text, text, LCG-random, records by b mod 4; 64 x 256 KiB in one call,
    several chains in the batch.  Every archive byte-identical to the reference's."""
    kinds = ["text", "text", "lcg", "records"]
    blocks = [corpus.block(kinds[b % 4], 1 << 18, corpus.BASE_SEED + b) for b in range(64)]
    arch = gpu.compress_blocks(blocks, "5")
    assert len({parse_block(a)["header"] for a in arch}) >= 2
    if "mixed64" not in _REF_CACHE:          # (this test runs three times -- alone and under both forced shapes: the reference's 20 s once)
        _REF_CACHE["mixed64"] = [ref.compress_block(d, "5") for d in blocks]
    for b, (a, r) in enumerate(zip(arch, _REF_CACHE["mixed64"])):
        assert a == r, b
    assert gpu.decompress(b"".join(arch)) == b"".join(b.tobytes() for b in blocks)


@pytest.mark.parametrize("shape", ["latency", "throughput"])
def test_both_shapes_of_the_pipelined_encoder(gpu, oracle, golden, ref, monkeypatch, shape):
    """The encoder has two shapes per chain (pipe_kernel.h: a lane per block / MIX, CM, MIX2 with a lane per bit position)
    and the engine picks one from the batch size.  Here each is FORCED (ZPAQ_AMD_PIPE_MODE) and the tests that cover the
    golden vectors, all nine component types, a batch that fills the GPU, several chains in one batch and a chain only
    hipRTC knows run again: whichever shape the engine picks in production has coded every one of these cases on the
    MI355X, bit-identical to the oracle / the reference."""
    monkeypatch.setenv("ZPAQ_AMD_PIPE_MODE", shape)
    test_all_nine_component_types(gpu, oracle, golden, 4)
    for idx in (1, 2):
        test_legacy_min_mid_max_models(gpu, golden, 4, idx)
    test_mixed_plans_in_one_batch_and_ragged_sizes(gpu, oracle)
    test_mixed_corpus_batch_against_the_reference(gpu, ref)
    # ... and both launch forms of the forced shape (the persistent launch is the default; the step kernels are what it
    # falls back to): the golden vectors through the step kernels
    monkeypatch.setenv("ZPAQ_AMD_PIPE_PERSIST", "0")
    test_encode_matches_oracle_and_golden(gpu, oracle, golden, 4)
    monkeypatch.delenv("ZPAQ_AMD_PIPE_PERSIST")
    note = __import__("ctypes").create_string_buffer(256)
    e = golden["config_cases"][0]
    assert gpu.lib().zpq_plan_kernel_kind3(gpu.Plan(bytes.fromhex(e["header"]))._h, 0, 4, note, 256) == 4


def _lcg_block(n, seed):
    """BASELINE.md's generator: x = x * 1664525 + 1013904223, byte = x >> 24, first byte after one step."""
    x, out = seed, bytearray(n)
    for i in range(n):
        x = (x * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = x >> 24
    return np.frombuffer(bytes(out), np.uint8)


def test_method_3_known_answer_and_preprocessing_levels(gpu, ref):
    """BASELINE configs[1]'s method: "3" = byte-aligned LZ77 through a suffix array on the host, then an ICM-ISSE
    chain (n = 2) on the GPU.  Known answer from BASELINE.md section 2; then levels 3 / 4 with every block-type hint
    (LZ77, BWT, E8E9 variants) against the reference, whole archives, and back through decompress."""
    a, = gpu.compress_blocks([_lcg_block(1 << 18, 12345)], "3")
    assert len(a) == 263462 and hashlib.sha1(a).hexdigest() == "cc9ec4cf41c67b408c8a1de88e00cfed3fcb07cf"
    r = np.random.default_rng(8)
    exe = r.integers(0, 256, 60000, dtype=np.uint8)
    exe[::37] = 0xE8
    exe[4::37] = 0
    blocks = [corpus.block("text", 70000, 31), corpus.block("records", 50000, 32), exe, corpus.block("lcg", 30000, 33),
              corpus.block("zeros", 40000, 34), corpus.block("text", 0, 35)]
    for m in ["3", "4", "3,128,1", "3,100,2", "3,30,0", "4,128,3", "4,30,2", "4,15,0", "4,240,1", "5,128,2"]:
        ours = gpu.compress_blocks([b.copy() for b in blocks], m)
        for b, a in zip(blocks, ours):
            assert a == ref.compress_block(b.copy(), m), m
        assert gpu.decompress(b"".join(ours)) == b"".join(b.tobytes() for b in blocks), m


SMALL_CHAIN_CFGS = [
    "comp 1 0 0 0 1\n  0 icm 12\nhcomp\n  *d=a halt\nend\n",
    "comp 2 0 0 0 2\n  0 icm 4\n  1 isse 4 0\nhcomp\n  b=a a=*d a<<= 4 a+=b *d=a d++ a<<= 3 a+=b *d=a halt\nend\n",
    "comp 2 3 0 0 2\n  0 icm 10\n  1 isse 10 0\nhcomp\n  c++ *c=a b=c a=0 d=0 hash b-- hash *d=a d++ b-- hash *d=a halt\nend\n",
    "comp 2 0 0 0 4\n  0 cm 9 255\n  1 icm 9\n  2 isse 10 1\n  3 isse 11 2\nhcomp\n  b=a *d=a d++ a=*d a<<= 8 a+=b *d=a d++ a<<= 2 a+=b *d=a d++ hash *d=a halt\nend\n",
]


def test_small_chains_in_the_latency_shape(gpu, oracle):
    """Round 6: the latency shape of a chain of a few unit wavefronts (BASELINE configs[1]'s n = 2) -- workgroups of 4, ISSE pairs
    unpacked, whole squash / stretch tables, ROW units with a lane per nibble and the table two bytes ahead, streams four bytes
    ahead in rings of fixed slots, HCOMP's small M array in LDS, the coder with a window per input byte (tests/test_emu.py has the
    same cases on the emulator).  Against the oracle: ragged, empty and one-byte blocks, zeros (every next row is the row being
    stored), incompressible bytes (the coder emits at nearly every bit; ~10^7 coded bits, some of which fail the fast form's
    test), several groups, output capacities around the coded length (status 3 exactly when it does not fit)."""
    assert gpu.lib().zpq_engine_count() >= 1
    blk = corpus.block("lcg", 1 << 18, corpus.BASE_SEED)
    h3, _, _ = gpu.method_to_header(gpu.expand_method("3", blk))
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65, 2000, 777, 5000, 513, 512, 511])]
    many = [b"\0" + corpus.block(kinds[i % 5], 100 + 131 * i, i).tobytes() for i in range(100)]
    long_ones = [corpus.block("lcg", 40000, 100 + i).tobytes() for i in range(33)]
    headers = [h3] + [gpu.assemble(c)[0] for c in SMALL_CHAIN_CFGS]
    for hdr in headers:
        plan = gpu.Plan(hdr)
        for inputs in ([ragged, many, long_ones] if hdr is h3 else [ragged]):
            got = gpu.encode_batch([plan] * len(inputs), inputs)
            assert gpu.lib().zpq_last_persistent() == 1
            for i, (g, d) in enumerate(zip(got, inputs)):
                assert g == oracle.encode(hdr, d), i
    plan = gpu.Plan(h3)
    data = [b"\0" + corpus.block("lcg", 400, 7).tobytes(), b"\0" + corpus.block("text", 400, 8).tobytes(), b"\0" + bytes(300)]
    want = [oracle.encode(h3, d) for d in data]
    for cap in (8, 40, 41, len(want[1]), len(want[0]) - 1, len(want[0]), len(want[0]) + 3, len(want[0]) + 39, len(want[0]) + 41):
        got, st, ol = gpu.encode_batch([plan] * 3, data, out_cap=[cap] * 3, check=False)
        for w, g, s_ in zip(want, got, st):
            assert (s_ == 0) == (len(w) <= cap), (cap, len(w), s_)
            assert g == w[:cap]


def test_sha1_on_the_device(gpu):
    """sha1_blocks_kernel against hashlib: lengths around every padding boundary, unaligned starts come from the
    PP-byte offset inside compress_blocks (whole-archive tests cover that), a few MiB-sized buffers."""
    import ctypes as C
    L = gpu.lib()
    rng = np.random.default_rng(4)
    lens = list(range(0, 130)) + [55, 56, 63, 64, 65, 119, 120, 127, 128, 1000, 4095, 4096, 65537, (1 << 20) + 3, 3 << 20]
    bufs = [rng.integers(0, 256, n, dtype=np.uint8) for n in lens]
    n = len(bufs)
    IA = (C.POINTER(C.c_uint8) * n)(*[b.ctypes.data_as(C.POINTER(C.c_uint8)) for b in bufs])
    LN = (C.c_uint32 * n)(*lens)
    out = np.zeros(20 * n, np.uint8)
    L.zpq_sha1_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    assert L.zpq_sha1_batch_device(IA, LN, n, out.ctypes.data) == 0, L.zpq_last_error()
    for i, b in enumerate(bufs):
        assert out[20 * i:20 * i + 20].tobytes() == hashlib.sha1(b.tobytes()).digest(), lens[i]


@pytest.mark.parametrize("mode", ["device", "host"])
def test_pcomp_post_processing_on_the_device(gpu, ref, monkeypatch, mode):
    """Reference archives of the LZ77 / BWT / E8E9 methods: the block's PCOMP program, translated to HIP, runs one
    lane per segment on the GPU (ZPAQ_AMD_PCOMP=device forces it even for this small batch); the host interpreter
    must give the same bytes."""
    monkeypatch.setenv("ZPAQ_AMD_PCOMP", mode)
    r = np.random.default_rng(8)
    exe = r.integers(0, 256, 50000, dtype=np.uint8)
    exe[::37] = 0xE8
    exe[4::37] = 0
    parts, stream = [], b""
    for kind, n, m in [("lcg", 40000, "3"), ("text", 60000, "3"), ("text", 50000, "3,128,1"), ("records", 30000, "4,30,0"),
                       ("text", 30000, "4,128,3"), ("pattern", 20000, "3,200,2"), ("text", 70000, "1"), ("records", 44000, "2"),
                       ("zeros", 10000, "1"), ("text", 0, "2"), ("text", 33000, "2,128,1")]:
        d = corpus.block(kind, n, 99)
        parts.append(d.tobytes())
        stream += ref.compress_block(d, m)
    for m in ("1,128,2", "3,100,2", "4,240,3", "x0,7", "x0,4"):
        parts.append(exe.tobytes())
        stream += ref.compress_block(exe.copy(), m)
    assert gpu.decompress(stream) == b"".join(parts)


def test_state_budget_forces_residency_waves_and_out_of_memory_recovers(gpu, oracle):
    """zpq_set_state_budget: a batch whose model state exceeds the budget is coded in several residency waves
    (engine.cpp, engine_code_host_on) -- same bytes as in one wave; a budget below ONE block's state fails with NOMEM
    ("Out of memory", the reference's message for it) and the next call, with the budget restored, works."""
    blocks = [corpus.block(["text", "records", "lcg"][i % 3], 20000 + 13 * i, 900 + i) for i in range(11)]
    whole = gpu.compress_blocks(blocks, "5")
    h = parse_block(whole[0])["header"]
    state = gpu.Plan(h).state_bytes
    try:
        gpu.set_state_budget(3 * state + (state >> 1))            # 3 blocks per wave -> 4 waves
        assert gpu.compress_blocks(blocks, "5") == whole
        assert gpu.decompress(b"".join(whole)) == b"".join(b.tobytes() for b in blocks)      # the decoder's waves as well
        gpu.set_state_budget(state // 2)
        with pytest.raises(gpu.ZpaqError) as ei:
            gpu.compress_blocks(blocks[:2], "5")
        assert ei.value.code == 1 and "ut of memory" in str(ei.value)
    finally:
        gpu.set_state_budget(0)
    assert gpu.compress_blocks(blocks, "5") == whole


def test_two_engines_shard_one_batch(gpu):
    """zpq_init(-1) with ZPAQ_AMD_DEVICES=0,0: two engines (two slots: own streams, arenas, loaded code objects) on the one
    GPU this box has -- the in-library sharding path of an N-GPU node (engine_code_host_now: contiguous block ranges,
    one host thread per engine).  Archives must come back in block order and equal the single-engine result."""
    import subprocess
    blocks = [corpus.block(["text", "lcg", "records", "zeros"][i % 4], 30000 + 7 * i, 300 + i) for i in range(37)]
    alone = gpu.compress_blocks(blocks, "5")
    code = ("import os, sys, hashlib\n"
            "os.environ['ZPAQ_AMD_DEVICES'] = '0,0'\n"
            "sys.path.insert(0, %r)\n"
            "import zpaq_amd as z\n"
            "from zpaq_amd import corpus\n"
            "z.init(-1)\n"
            "blocks = [corpus.block(['text', 'lcg', 'records', 'zeros'][i %% 4], 30000 + 7 * i, 300 + i) for i in range(37)]\n"
            "arch = z.compress_blocks(blocks, '5')\n"
            "assert z.decompress(b''.join(arch)) == b''.join(b.tobytes() for b in blocks)\n"
            "print('DIGEST', hashlib.sha1(b''.join(arch)).hexdigest(), len(arch))\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "DIGEST %s 37" % hashlib.sha1(b"".join(alone)).hexdigest() in r.stdout


def test_eight_engines_on_one_gpu_shard_an_uneven_batch(gpu):
    """The in-library path of an 8-GPU node (zpq_init(-1): one engine, one host thread and one contiguous block range per
    configured device) with the one GPU of this box named eight times: 1027 blocks -- not a multiple of 8, so the ranges are
    uneven -- of four kinds (two chains) and ragged lengths, coded by eight engines side by side (their persistent launches take turns on
    the shared device).  Archives in block order, identical to the single-engine result, and they decode back."""
    import subprocess
    code = ("import os, sys, hashlib\n"
            "sys.path.insert(0, %r)\n"
            "import zpaq_amd as z\n"
            "from zpaq_amd import corpus\n"
            "n = 1027\n"
            # (the records blocks share one seed: one chain of 27 components that the build knows beside the standard one -- 257
            #  'pattern' blocks with their own detected periods were 38 chains for hipRTC and two and a half minutes)
            "blocks = [corpus.block(['text', 'lcg', 'records', 'zeros'][i %% 4], 1500 + (i * 37) %% 2500, 902 if i %% 4 == 2 else 900 + i) for i in range(n)]\n"
            "z.init(-1 if os.environ.get('ZPAQ_AMD_DEVICES') else 0)\n"
            "arch = z.compress_blocks(blocks, '5')\n"
            "assert z.decompress(b''.join(arch[:40])) == b''.join(b.tobytes() for b in blocks[:40])\n"
            "print('DIGEST', hashlib.sha1(b''.join(arch)).hexdigest(), len(arch), z.lib().zpq_engine_count())\n" % ROOT)
    outs = []
    gpu.shutdown()           # (this process's engine may hold most of the HBM from earlier tests: the children get the device)
    for devs in (None, "0,0,0,0,0,0,0,0"):
        env = {k: v for k, v in os.environ.items() if k != "ZPAQ_AMD_DEVICES"}
        if devs:
            env["ZPAQ_AMD_DEVICES"] = devs
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][0].split())
    gpu.init(0)
    assert outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2] == "1027"
    assert outs[1][3] == "8", outs


def test_persistent_launch_gives_up_and_the_step_kernels_take_over(gpu, oracle, monkeypatch):
    """The persistent encoder launch (device/pipe_persist.h) spins on progress counters, so it carries a watchdog: a poller
    that sees no progress for ZPAQ_AMD_PERSIST_TIMEOUT_MS raises the abort word, every unit exits, and the engine codes the
    batch again with the step kernels on re-initialised arenas.  With a 1 ms limit the coder (which waits for the whole
    pipeline to fill) gives up for certain: the archives must still be the oracle's, byte for byte."""
    blocks = [corpus.block(["text", "records", "lcg"][i % 3], 150_000 + 911 * i, 70 + i) for i in range(40)]
    want = gpu.compress_blocks(blocks, "5")
    assert gpu.lib().zpq_last_persistent() == 1
    monkeypatch.setenv("ZPAQ_AMD_PERSIST_TIMEOUT_MS", "1")
    got = gpu.compress_blocks(blocks, "5")
    assert gpu.lib().zpq_last_persistent() == 0, "the watchdog did not fire"
    monkeypatch.delenv("ZPAQ_AMD_PERSIST_TIMEOUT_MS")
    assert got == want
    again = gpu.compress_blocks(blocks, "5")
    assert gpu.lib().zpq_last_persistent() == 1 and again == want
    monkeypatch.setenv("ZPAQ_AMD_PIPE_PERSIST", "0")
    assert gpu.compress_blocks(blocks, "5") == want and gpu.lib().zpq_last_persistent() == 0
    f = parse_block(want[0])
    coded = oracle.encode(f["header"], b"\0" + blocks[0].tobytes())
    assert want[0][f["payload_start"]:f["payload_start"] + len(coded)] == coded


def test_foreign_kernel_on_the_device_makes_the_persistent_launch_step_aside(gpu, tmp_path):
    """The persistent launch needs all its workgroups resident together; what the library cannot know about -- a second
    process's kernel, here tests/cpp/gpu_hog.hip holding 200 of the 256 compute units with 140 KiB of LDS each -- leaves
    some of them in the queue.  The arrival handshake (pipe_persist.h pipe_arrived) notices within ZPAQ_AMD_PERSIST_ARRIVE_MS
    (20 ms) that the count has stopped short, nothing has been touched, and the step kernels code the batch beside the foreign
    kernel: same archives, given up in well under 100 ms instead of after the 3 s watchdog."""
    import ctypes as C
    import subprocess
    import time
    hog = str(tmp_path / "gpu_hog")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", os.path.join(ROOT, "tests", "cpp", "gpu_hog.hip"), "-o", hog],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    blocks = [corpus.block(["text", "records"][i % 2], 32768 + 64 * i, 300 + i) for i in range(256)]
    L = gpu.lib()
    L.zpq_last_persist_abort_ms.restype = C.c_double
    want = gpu.compress_blocks(blocks, "5")
    assert L.zpq_last_persistent() == 1 and L.zpq_last_persist_abort_ms() == 0.0
    p = subprocess.Popen([hog, "200", "6000"], stdout=subprocess.PIPE, text=True)
    try:
        assert "holding" in p.stdout.readline()
        time.sleep(0.3)                                  # (the foreign workgroups are on their compute units)
        t0 = time.time()
        got = gpu.compress_blocks(blocks, "5")
        wall = time.time() - t0
        gave_up_ms = L.zpq_last_persist_abort_ms()
        assert p.poll() is None, "the foreign kernel ended before the batch was coded: nothing was tested"
        assert L.zpq_last_persistent() == 0, "the persistent launch ran although 200 compute units were taken"
        assert 0.0 < gave_up_ms < 100.0, gave_up_ms
        assert got == want
        print(f"given up after {gave_up_ms:.1f} ms, batch coded by the step kernels in {wall:.2f} s beside the foreign kernel")
    finally:
        p.kill()
        p.wait()
    assert gpu.compress_blocks(blocks, "5") == want and L.zpq_last_persistent() == 1


def test_the_librarys_own_hashing_kernel_does_not_make_the_persistent_launch_step_aside(gpu):
    """zpq_compress_blocks hashes the blocks on the device beside the coder (sha1_blocks_kernel: a lane per block, 46 ms for
    1 MiB blocks).  A batch that needs EVERY compute unit for its persistent launch must not find a few of them held by that
    kernel for longer than the arrival handshake waits (round 6, call 9: the API leg of the bench fell to the step kernels,
    267 instead of 357 MB/s): the engine waits for the hashing before the grid arrives."""
    import ctypes as C
    base = np.random.default_rng(5000).integers(0, 256, 1 << 20, dtype=np.uint8)
    blocks = [np.roll(base, 977 * i) ^ np.uint8(i & 255) for i in range(1024)]         # (1024 different incompressible blocks, cheaply)
    L = gpu.lib()
    L.zpq_last_persist_abort_ms.restype = C.c_double
    arch = gpu.compress_blocks(blocks, "5")
    assert L.zpq_last_persistent() == 1 and L.zpq_last_persist_abort_ms() == 0.0, L.zpq_last_persist_abort_ms()
    for i in (0, 511, 1023):
        assert hashlib.sha1(blocks[i].tobytes()).digest() in arch[i]          # the segment's SHA-1 trailer (hashed on the device)
    assert gpu.decompress(arch[1023]) == blocks[1023].tobytes()


def test_measured_and_shelved_forms_stay_bit_exact(gpu, tmp_path):
    """Two forms of round 6 that were built, measured slower on the MI355X and left off by default: MIX
    weight rows packed as 24-bit quads with the first rows of a small table in LDS (ZPAQ_AMD_MIX_PACKED=1), and the lockstep
    decoder with a tail wavefront (ZPAQ_AMD_TEAM_TAIL=1).  The generator reads its knobs once per process: a child codes
    blocks of the headline's chain with both on (code objects from the prebuilt cache, hipRTC otherwise) -- the coded streams
    must be the default's byte for byte, and the tail form must decode them."""
    import subprocess
    blocks = [corpus.block("text", 1 << 20, 900 + i) for i in range(12)]
    hdr = gpu.method_to_header(gpu.expand_method("5", blocks[0]))[0]
    plan = gpu.Plan(hdr)
    want = gpu.encode_batch([plan] * len(blocks), [b"\0" + b.tobytes() for b in blocks])
    digest = hashlib.sha1(b"".join(want)).hexdigest()
    code = ("import sys, hashlib; sys.path.insert(0, %r)\n"
            "import zpaq_amd as z\nfrom zpaq_amd import corpus, prebuild\n"
            "blocks = [corpus.block('text', 1 << 20, 900 + i) for i in range(12)]\n"
            "hdr = z.method_to_header(z.expand_method('5', blocks[0]))[0]\n"
            "assert 'MIX_PACKED[2] = {1,1}' in prebuild.pipe_source_and_key(hdr, 0)[0] and '__launch_bounds__(448)' in prebuild.team_source_and_key(hdr)[0]\n"
            "z.init(0)\nplan = z.Plan(hdr)\n"
            "coded = z.encode_batch([plan] * 12, [b'\\0' + b.tobytes() for b in blocks])\n"
            "print('PERSIST', z.lib().zpq_last_persistent())\n"
            "print('DIGEST', hashlib.sha1(b''.join(coded)).hexdigest())\n"
            "z.set_kernel(6)\n"
            "back = z.decode_batch([plan] * 12, [c + b'\\0\\0\\0\\0' for c in coded], [40001] * 12)\n"
            "print('DECODED', all(d == (b'\\0' + blocks[i].tobytes())[:40001] for i, (d, _) in enumerate(back)))\n" % ROOT)
    env = dict(os.environ, ZPAQ_AMD_MIX_PACKED="1", ZPAQ_AMD_TEAM_TAIL="1", ZPAQ_AMD_PIPE_MODE="throughput")
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "PERSIST 1" in r.stdout and f"DIGEST {digest}" in r.stdout and "DECODED True" in r.stdout, r.stdout[-2000:]


def test_input_tail_copied_behind_the_first_steps(gpu, monkeypatch):
    """A batch of 64 or more equally long blocks of 512 KiB or more from pinned staging has only the first 64 KiB of every block
    on the device when the step kernels start; the rest follows behind the first steps (engine.cpp LateInput: step s reads
    input below (s + 1) x chunk only -- PipeLane::byte_at never looks ahead of its chunk).  Same archives with the split off."""
    blocks = [corpus.block(["text", "lcg"][i % 2], 512 << 10, 40 + i) for i in range(64)]
    monkeypatch.setenv("ZPAQ_AMD_PIPE_PERSIST", "0")         # (the persistent launch waits for the whole input)
    split = gpu.compress_blocks(blocks, "5")
    monkeypatch.setenv("ZPAQ_AMD_SPLIT_COPY", "0")
    whole = gpu.compress_blocks(blocks, "5")
    assert split == whole
    monkeypatch.delenv("ZPAQ_AMD_PIPE_PERSIST")
    assert gpu.compress_blocks(blocks, "5") == whole


def test_suffix_arrays_on_the_device(gpu, ref):
    """The sort inside the byte-aligned LZ77 and BWT pre-processors (reference: divsufsort, libzpaq.cpp:4658-6434), for a
    whole batch in one device call (device/sa_kernels.hip: prefix doubling, one radix sort per round over every block).
    A suffix array is canonical: the device's must equal the host sorter's, entry for entry -- text, random bytes, a block
    of zeros (log2(n) rounds), periodic records, ragged and tiny buffers -- and the archives of methods that sort (3: LZ77
    through a suffix array; 3 with the text hint: BWT; 2: bit-packed LZ77) made from a batch large enough to take the
    device path must be the reference's, byte for byte."""
    import ctypes as C
    L = gpu.lib()
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    bufs = [corpus.block(kinds[i % 5], n, 70 + i) for i, n in enumerate([70000, 65536, 40000, 50001, 30000, 1, 2, 3, 255, 256, 257, 1000, 99999])]
    n = len(bufs)
    outs = [np.empty(max(b.size, 1), np.uint32) for b in bufs]
    u8p, u32p = C.POINTER(C.c_ubyte), C.POINTER(C.c_uint32)
    IA = (u8p * n)(*[b.ctypes.data_as(u8p) for b in bufs])
    LN = (C.c_uint32 * n)(*[b.size for b in bufs])
    OA = (u32p * n)(*[o.ctypes.data_as(u32p) for o in outs])
    L.zpq_suffix_arrays_device.argtypes = [C.POINTER(u8p), C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(u32p)]
    L.zpq_suffix_array_host.argtypes = [u8p, C.c_uint32, u32p]
    assert L.zpq_suffix_arrays_device(IA, LN, n, OA) == 0, L.zpq_last_error().decode()
    for b, o in zip(bufs, outs):
        want = np.empty(max(b.size, 1), np.uint32)
        assert L.zpq_suffix_array_host(b.ctypes.data_as(u8p), b.size, want.ctypes.data_as(u32p)) == 0
        assert (o[:b.size] == want[:b.size]).all(), (b.size, int(b[0]))
        assert (o[:b.size].astype(np.int64) == ref.divsufsort(b.tobytes()).astype(np.int64)).all(), b.size     # ... and the reference's own
    # through compressBlock's own methods, batches that take the device path (4 or more sorting blocks, 1 MiB or more)
    blocks = [corpus.block(kinds[i % 4], 150000 + 1111 * i, 500 + i) for i in range(12)]
    ph = (C.c_double * 8)()
    for method in ("3", "3,128,1", "2"):        # byte-aligned LZ77 + ICM/ISSE, BWT + ICM/ISSE, bit-packed LZ77 (no model)
        arch = gpu.compress_blocks(blocks, method)
        L.zpq_last_api_timing(ph)
        assert int(ph[7]) == len(blocks), (method, ph[7])          # every block's suffix array came from the device
        for d, a in zip(blocks, arch):
            assert a == ref.compress_block(d, method), (method, d.size)
    assert gpu.decompress(b"".join(arch)) == b"".join(b.tobytes() for b in blocks)


def test_lz77_parse_and_bwt_on_the_device(gpu, ref, monkeypatch):
    """The pre-processors behind the sort on the device (device/lz77_kernel.h; reference: LZBuffer::fill with a suffix array,
    libzpaq.cpp:6693-6757, and divbwt's output): for a whole batch the stream zpq_preprocess_blocks_device returns -- sort,
    parse and BWT on the GPU, LZBuffer's codes written by the host from the list of matches -- must be the host's, byte for
    byte: both code levels, look-aheads 0..3, E8E9 in front, text / random / zeros / records / patterns, ragged, tiny and
    empty buffers, a 1 MiB block (eight windows of the inverse array).  Then the archives of compressBlock's sorting methods,
    from batches that take the device path, against the reference -- with the parse on the device and (knob) on the host."""
    import ctypes as C
    L = gpu.lib()
    u8p = C.POINTER(C.c_ubyte)
    L.zpq_preprocess_blocks_device.argtypes = [C.c_char_p, C.POINTER(u8p), C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(u8p), C.POINTER(C.c_size_t),
                                               C.POINTER(C.c_size_t)]
    L.zpq_preprocess_block.argtypes = [C.c_char_p, u8p, C.c_uint32, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    sizes = [70000, 65536, 40000, 50001, 30000, 1, 2, 3, 0, 255, 256, 257, 1000, 99999, 1 << 20]
    for xm in ("x0,2,5,0,7,21,1c0,0,511", "x0,1,4,0,3,21,1", "x0,2,12,0,7,21,1c0,0,511i2", "x0,6,5,0,2,21,0c0,0,511", "x0,2,4,0,7,21,3c0,0,511",
               "x0,5,5,0,5,21,2", "x0,3ci1", "x0,7ci1"):
        src = [corpus.block(kinds[i % 5], n, 70 + i) if n else np.zeros(0, np.uint8) for i, n in enumerate(sizes)]
        dev_in = [np.concatenate([b, np.zeros(8, np.uint8)]) for b in src]        # (copies: E8E9 works in place)
        n = len(src)
        outs = [np.empty(b.size + b.size // 2 + 4096, np.uint8) for b in src]
        IA = (u8p * n)(*[b.ctypes.data_as(u8p) for b in dev_in])
        LN = (C.c_uint32 * n)(*[b.size for b in src])
        OA = (u8p * n)(*[o.ctypes.data_as(u8p) for o in outs])
        CP = (C.c_size_t * n)(*[o.size for o in outs])
        OL = (C.c_size_t * n)()
        assert L.zpq_preprocess_blocks_device(xm.encode(), IA, LN, n, OA, CP, OL) == 0, (xm, L.zpq_last_error().decode())
        for k, b in enumerate(src):
            host_in = b.copy() if b.size else np.zeros(1, np.uint8)
            want = np.empty(outs[k].size, np.uint8)
            wl = C.c_size_t(0)
            assert L.zpq_preprocess_block(xm.encode(), host_in.ctypes.data_as(u8p), b.size, want.ctypes.data_as(u8p), want.size, C.byref(wl)) == 0
            assert OL[k] == wl.value and (outs[k][:wl.value] == want[:wl.value]).all(), (xm, k, b.size, OL[k], wl.value)
    blocks = [corpus.block(kinds[i % 4], 150000 + 1111 * i, 500 + i) for i in range(12)]
    ph = (C.c_double * 8)()
    for knob in ("1", "0"):
        monkeypatch.setenv("ZPAQ_AMD_DEVICE_PARSE", knob)
        for method in ("3", "3,128,1", "2", "x0,6,5,0,7,21,1c0,0,511", "x0,7ci1"):
            arch = gpu.compress_blocks([b.copy() for b in blocks], method)
            L.zpq_last_api_timing(ph)
            assert int(ph[7]) == len(blocks), (method, ph[7])
            for d, a in zip(blocks, arch):
                assert a == ref.compress_block(d.copy(), method), (knob, method, d.size)
        assert gpu.decompress(b"".join(arch)) == b"".join(b.tobytes() for b in blocks)
