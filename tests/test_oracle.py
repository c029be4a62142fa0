"""CPU: pins the plain-C oracle against the reference's known answers, the
committed golden vectors (made by the reference itself) and, where oracle/_ref
is present, live differential runs."""
import hashlib

import numpy as np
import pytest

from conftest import b64, gen_input
from oracle.oracle_py import parse_block
from zpaq_amd import corpus


def test_table_checksums(oracle):
    # stsum / sqsum asserted by the reference in DEBUG builds (libzpaq.cpp:1759-1760)
    assert oracle.tables_ok()
    st = oracle.table("stretch").astype(np.int64)
    sq = oracle.table("squash").astype(np.int64)
    assert sq[0] == 0 and sq[4095] == 32767 and sq[2048] == 16384
    assert (st[:16384] == -st[32767:16383:-1]).all()
    assert oracle.table("dt")[0] == (1 << 17) // 3 * 2 and oracle.table("dt2k")[1] == 2048


def test_state_table_matches_reference(oracle, ref):
    assert (oracle.table("state") == ref.state_table()).all()


def _check_entry(oracle, e, archive=None):
    hdr = bytes.fromhex(e["header"])
    d = gen_input(e)
    ps = e["payload_start"]
    if hdr[6] == 0:
        return  # stored blocks have no modelled payload
    coded = oracle.encode(hdr, b"\0" + d.tobytes())
    if archive is not None:
        assert archive[ps:ps + len(coded)] == coded
        assert archive[ps + len(coded):ps + len(coded) + 4] == b"\0\0\0\0"
        dec, used = oracle.decode(hdr, archive[ps:], len(d) + 16)
        assert dec == b"\0" + d.tobytes() and used == len(coded) + 4
    # tag(13)+zPQ..(5)+header .. payload_start is the prologue; epilogue = 4 zeros + 253 sha1 + 255
    total = ps + len(coded) + 4 + 21 + 1
    assert total == e["len"]
    return coded


def test_oracle_vs_golden_small_archives(oracle, golden):
    n = 0
    for e in golden["method_cases"]:
        if "archive_b64" in e and e["n"] <= 65536:
            _check_entry(oracle, e, b64(e))
            n += 1
    assert n >= 10


def test_oracle_vs_golden_sha1(oracle, golden):
    """Rebuild whole archives around the oracle's coded stream and compare SHA-1 with the reference's."""
    from oracle.oracle_py import TAG
    n = 0
    for e in golden["method_cases"]:
        if e["n"] > 70000 or e["method"] == "0":
            continue
        hdr = bytes.fromhex(e["header"])
        d = gen_input(e)
        coded = oracle.encode(hdr, b"\0" + d.tobytes())
        comment = str(e["n"]) + ((" " + e["comment"]) if e["comment"] else "")
        a = (TAG + b"zPQ" + bytes([1 + (hdr[6] == 0), 1]) + hdr + b"\x01" + (e["filename"] or "").encode() + b"\0"
             + comment.encode() + b"\0\0" + coded + b"\0\0\0\0\xfd" + hashlib.sha1(d.tobytes()).digest() + b"\xff")
        assert len(a) == e["len"] and hashlib.sha1(a).hexdigest() == e["sha1"], (e["kind"], e["n"], e["method"])
        n += 1
    assert n >= 40


def test_oracle_all_nine_component_types(oracle, golden):
    e = golden["config_cases"][0]
    _check_entry(oracle, e, b64(e))


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_oracle_legacy_min_mid_max(oracle, golden, idx):
    e = golden["level_cases"][idx]
    _check_entry(oracle, e, b64(e))


def test_oracle_memory_matches_reference_report(oracle, golden):
    for e in golden["method_cases"]:
        hdr = bytes.fromhex(e["header"])
        if hdr[6]:
            assert oracle.memory(hdr) == e["memory"]


def test_known_answers_live(ref, golden):
    """BASELINE.md §2 known answers reproduce with the reference built here (plumbing config C1 included)."""
    for e in golden["known_answers"]:
        if e["n"] > 300000:
            continue
        a = ref.compress_block(corpus.block(e["kind"], e["n"], e["seed"]), e["method"])
        assert len(a) == e["len"] and hashlib.sha1(a).hexdigest() == e["sha1"]


def test_config1_plumbing_cpu_reference(ref):
    """BASELINE config[0]: libzpaq::compress -m1 on 16 x 64 KiB zero buffers through the CPU reference."""
    z = np.zeros(65536, np.uint8)
    for _ in range(16):
        a = ref.compress(z, "1")
        assert len(a) == 394 and hashlib.sha1(a).hexdigest() == "20fb8eb50acb4e41a6a6d455e388ffde0c7ebe4a"
        assert ref.decompress(a, 65536) == z.tobytes()


def test_oracle_live_differential(oracle, ref):
    """Random sizes / kinds / hints against the live reference."""
    rng = np.random.RandomState(5)
    for it in range(12):
        kind = ["text", "lcg", "records", "pattern", "zeros"][it % 5]
        n = int(rng.randint(1, 30000))
        method = ["5", "4", "5,100,1", "4,200,1"][it % 4]
        d = corpus.block(kind, n, 1000 + it)
        a = ref.compress_block(d, method)
        f = parse_block(a)
        coded = oracle.encode(f["header"], b"\0" + d.tobytes())
        ps = f["payload_start"]
        assert a[ps:ps + len(coded) + 4] == coded + b"\0\0\0\0"
        dec, _ = oracle.decode(f["header"], a[ps:], n + 16)
        assert dec[1:] == d.tobytes()


def test_oracle_decoder_detects_corruption(oracle, golden):
    e = [x for x in golden["method_cases"] if x["kind"] == "text" and x["n"] == 777 and x["method"] == "5"][0]
    a = bytearray(b64(e))
    hdr = bytes.fromhex(e["header"])
    ps = e["payload_start"]
    a[ps + 10] ^= 0x55
    try:
        dec, _ = oracle.decode(hdr, bytes(a[ps:]), 2000)
        assert dec[1:] != gen_input(e).tobytes()
    except RuntimeError:
        pass


def test_oracle_random_hcomp_programs(oracle, golden):
    """Random ZPAQL programs (all operand kinds, swaps, hash/hashd, R, div/mod by zero, shifts,
    IF/ELSE, IFL/ELSEL long jumps, DO loops): pins the oracle's HCOMP VM to the reference's."""
    assert len(golden["vm_cases"]) >= 10
    for e in golden["vm_cases"]:
        _check_entry(oracle, e, b64(e))


def test_oracle_seventy_components(oracle, golden):
    e = [c for c in golden["config_cases"] if c["name"] == "seventy_components"][0]
    assert bytes.fromhex(e["header"])[6] == 70
    _check_entry(oracle, e, b64(e))


def test_oracle_equals_the_reference_on_random_models(oracle, ref, zlib_):
    """Random MODELS, not only the fixed ones: 1-24 components of every type with random sizes, inputs, rates and masks and a
    random loop-free HCOMP program (tests/fuzz_emu.py's generator), compiled and run by the reference (oracle/_ref) and by
    the oracle on text, records and noise: same header, same coded bytes, and the oracle's decoder returns the input.
    tests/fuzz_emu.py puts the device code (under the wavefront emulator) against the oracle on the same family."""
    import os
    import random
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_emu
    rng = random.Random(20260926)
    done = 0
    while done < 150:
        cfg = fuzz_emu.random_model(rng, rng.random() < 0.3)
        try:
            header, _ = zlib_.assemble(cfg)
            zlib_.Plan(header)
        except zlib_.ZpaqError:
            continue
        for kind, n in (("text", 3000), ("records", 2000), ("lcg", 500)):
            d = corpus.block(kind, n, rng.randrange(1 << 20)).tobytes()
            a = ref.compress_config(d, cfg, None, "f", None, False)
            f = parse_block(a)
            ps = f["payload_start"]
            assert f["header"] == header, cfg
            try:
                coded = oracle.encode(header, b"\0" + d)
            except RuntimeError:          # a model that predicts this badly expands past the wrapper's buffer
                break
            assert a[ps:ps + len(coded) + 4] == coded + b"\0\0\0\0", (kind, cfg)
            assert oracle.decode(header, coded + b"\0\0\0\0", n + 1)[0] == b"\0" + d, (kind, cfg)
        else:
            done += 1
