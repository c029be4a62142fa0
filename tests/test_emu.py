"""CPU (-m "not gpu"): the DEVICE source of the specialised coder, executed on the host by the wavefront
emulator under tests/emu (64 lanes as fibers; readlane / DPP / shuffles / LDS modelled after the gfx9 ISA), must
produce the oracle's bytes.  The GPU parity tests remain the proof for the hardware; this catches logic errors in
spec_kernel.h and in the generator before a GPU is involved, for both workgroup shapes."""
import os
import sys

import numpy as np
import pytest

from conftest import b64, gen_input

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
import emu  # noqa: E402

from zpaq_amd import corpus  # noqa: E402

DEEP_ISSE = "x0,0ci1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1m"


def _check(oracle, header, datas, waves):
    inputs = [b"\0" + bytes(d) for d in datas]
    enc = emu.run(header, inputs, waves=waves)
    for inp, (coded, status, consumed) in zip(inputs, enc):
        assert status == 0 and consumed == len(inp)
        assert coded == oracle.encode(header, inp)
    dec = emu.run(header, [c + b"\0\0\0\0" for c, _, _ in enc], decode=True, waves=waves,
                  out_cap=max(len(x) for x in inputs))
    for inp, (c, _, _), (plain, status, consumed) in zip(inputs, enc, dec):
        # a block that fills its capacity exactly stops before the end-of-stream marker
        assert status == 0 and plain == inp[:len(plain)] and len(plain) == len(inp)


def _ragged(n):
    return [corpus.block("text", n, 5).tobytes(), corpus.block("records", n + 37, 6).tobytes(),
            corpus.block("lcg", n // 2, 7).tobytes(), corpus.block("zeros", n // 3, 8).tobytes(), b""]


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("method", ["5", "4", "5,128,1"])
def test_standard_chains_in_both_shapes(zlib_, oracle, method, waves):
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    header, _, _ = zlib_.method_to_header(zlib_.expand_method(method, blk))
    _check(oracle, header, _ragged(700), waves)


@pytest.mark.parametrize("waves", [4, 8])
def test_side_tables_in_the_arena(zlib_, oracle, waves):
    """17 ISSE + 1 ICM do not fit a block's LDS in either shape: the last ones stay in the arena
    (fetched one bit ahead with register forwarding in the 4-block shape)."""
    header, _, _ = zlib_.method_to_header(DEEP_ISSE)
    _check(oracle, header, _ragged(900), waves)


def test_all_nine_component_types_and_legacy_models(zlib_, oracle, golden):
    for e in [golden["config_cases"][0]] + golden["level_cases"]:
        header = bytes.fromhex(e["header"])
        _check(oracle, header, [gen_input(e).tobytes()[:1500]], 4)


def test_random_hcomp_programs(zlib_, oracle, golden):
    """The generator's HCOMP -> C++ translation (jumps, all operand modes) against the oracle's interpreter."""
    for e in golden["vm_cases"][:6]:
        header = bytes.fromhex(e["header"])
        if header[6] == 0:
            continue
        _check(oracle, header, [corpus.block("text", 400, 31).tobytes(), corpus.block("lcg", 300, 32).tobytes()], 4)


def test_every_golden_chain(zlib_, oracle, golden):
    """Every distinct chain among the golden vectors (periodic models from level-5 period detection included: up to
    31 components), a prefix of the vector's own input: emulator == oracle, and the oracle's bytes are a prefix of
    the reference archive's payload when the whole input was coded."""
    seen = set()
    for e in golden["method_cases"]:
        header = bytes.fromhex(e["header"])
        if header[6] == 0 or header in seen:
            continue
        seen.add(header)
        data = gen_input(e).tobytes()
        if len(data) < 64:
            data = corpus.block(e["kind"] if e["kind"] != "mixed" else "records", 600, 3).tobytes()
        _check(oracle, header, [data[:600]], 4)
    assert len(seen) >= 8


def test_decoder_contract_on_bad_and_partial_streams(zlib_, oracle):
    """Decoder::decode's error rules and Decompresser::decompress(n)'s "first n bytes": a truncated stream ends with
    status 6 (EOF) or 2 (corrupt), never with output past what was coded; a capacity smaller than the block returns
    exactly that prefix with consumed = 0; garbage does not crash and does not reproduce the data."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    header, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    d = b"\0" + corpus.block("text", 1500, 77).tobytes()
    c = oracle.encode(header, d)
    good = c + b"\0\0\0\0"
    rng = np.random.default_rng(5)
    garbage = rng.integers(0, 256, 400, dtype=np.uint8).tobytes()
    res = emu.run(header, [c[:len(c) // 2], good, garbage, good], decode=True, waves=4, out_cap=len(d) + 8)
    (t_out, t_st, _), (g_out, g_st, g_used), (x_out, x_st, _), (g2_out, g2_st, _) = res
    assert t_st in (6, 2) and d.startswith(t_out[:len(t_out) - 1] if t_out else b"")
    assert g_st == 0 and g_out == d and g_used == len(c) + 4
    assert x_st in (0, 2, 6) and x_out != d
    assert g2_st == 0 and g2_out == d
    (p_out, p_st, p_used), = emu.run(header, [good], decode=True, waves=4, out_cap=701)
    assert p_st == 0 and p_used == 0 and p_out == d[:701]


# ---------------------------------------------------------------------------------------------------------
# The decoder with two blocks per wavefront (zpaq_amd/csrc/device/spec_dual_kernel.h)

def _dual_check(oracle, header, datas):
    inputs = [b"\0" + bytes(d) for d in datas]
    coded = [oracle.encode(header, i) for i in inputs]
    dec = emu.run(header, [c + b"\0\0\0\0" for c in coded], decode=True, out_cap=max(len(x) for x in inputs), dual=True)
    for inp, c, (plain, status, consumed) in zip(inputs, coded, dec):
        assert status == 0 and plain == inp
        assert consumed in (0, len(c) + 4)        # 0: the block filled the capacity exactly and stopped before the marker


@pytest.mark.parametrize("method", ["5", DEEP_ISSE])
def test_two_blocks_per_wavefront_decoder(zlib_, oracle, method):
    """Standard chains, the chain whose side tables do not fit the LDS, ragged lengths, an empty block, an odd number of
    blocks (the last wavefront has one block only) and more than one workgroup."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    header = zlib_.method_to_header(method if method.startswith("x") else zlib_.expand_method(method, blk))[0]
    _dual_check(oracle, header, _ragged(600) + [corpus.block("text", n, n).tobytes() for n in (1, 64, 333, 600)])


def test_two_blocks_per_wavefront_decoder_on_every_component_type(zlib_, oracle, golden):
    for e in [golden["config_cases"][0]] + golden["level_cases"][1:] + golden["vm_cases"][:1]:
        header = bytes.fromhex(e["header"])
        if header[6] > 32:
            continue
        d = gen_input(e).tobytes()[:1200]
        _dual_check(oracle, header, [d, d[:700], d[:1]])


def test_two_blocks_per_wavefront_decoder_contract_on_bad_streams(zlib_, oracle):
    """The same answers as the one-block kernel gives on a truncated stream, on garbage and on a capacity below the block."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    header, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    d = b"\0" + corpus.block("text", 1500, 77).tobytes()
    c = oracle.encode(header, d)
    good = c + b"\0\0\0\0"
    garbage = np.random.default_rng(5).integers(0, 256, 400, dtype=np.uint8).tobytes()
    streams = [c[:len(c) // 2], good, garbage, good]
    one = emu.run(header, streams, decode=True, waves=8, out_cap=len(d) + 8)
    two = emu.run(header, streams, decode=True, out_cap=len(d) + 8, dual=True)
    assert one == two
    assert two[1][1] == 0 and two[1][0] == d and two[1][2] == len(c) + 4
    assert emu.run(header, [good], decode=True, out_cap=701, dual=True) == emu.run(header, [good], decode=True, waves=8, out_cap=701)


# ---------------------------------------------------------------------------------------------------------
# The lockstep decoder (zpaq_amd/csrc/device/spec_team_kernel.h): row wavefronts + mixer wavefronts, workgroup barriers

def _team_check(oracle, header, datas):
    inputs = [b"\0" + bytes(d) for d in datas]
    coded = [oracle.encode(header, i) for i in inputs]
    dec = emu.run(header, [c + b"\0\0\0\0" for c in coded], decode=True, out_cap=max(len(x) for x in inputs), team=True)
    for inp, c, (plain, status, consumed) in zip(inputs, coded, dec):
        assert status == 0 and plain == inp
        assert consumed in (0, len(c) + 4)


@pytest.mark.parametrize("method", ["5", DEEP_ISSE])
def test_lockstep_decoder(zlib_, oracle, method):
    """-m5 (16 ICM / ISSE components: 16 row lanes per block, 384 threads) and the chain of 18 whose side tables do not fit
    the LDS (32 row lanes per block, 512 threads); ragged lengths, an empty block, 9 blocks = a full workgroup and one
    with a single block."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    header = zlib_.method_to_header(method if method.startswith("x") else zlib_.expand_method(method, blk))[0]
    _team_check(oracle, header, _ragged(500) + [corpus.block("text", n, n).tobytes() for n in (1, 64, 333, 500)])


def test_lockstep_decoder_on_every_component_type_and_bad_streams(zlib_, oracle, golden):
    for e in [golden["config_cases"][0]] + golden["level_cases"][1:] + golden["vm_cases"][:1]:
        header = bytes.fromhex(e["header"])
        d = gen_input(e).tobytes()[:1000]
        _team_check(oracle, header, [d, d[:600], d[:1]])
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    header, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    d = b"\0" + corpus.block("text", 1200, 77).tobytes()
    c = oracle.encode(header, d)
    good = c + b"\0\0\0\0"
    garbage = np.random.default_rng(5).integers(0, 256, 400, dtype=np.uint8).tobytes()
    streams = [c[:len(c) // 2], good, garbage, good]
    one = emu.run(header, streams, decode=True, waves=8, out_cap=len(d) + 8)
    team = emu.run(header, streams, decode=True, out_cap=len(d) + 8, team=True)
    assert one == team
    assert team[1][1] == 0 and team[1][0] == d and team[1][2] == len(c) + 4
    assert emu.run(header, [good], decode=True, out_cap=701, team=True) == emu.run(header, [good], decode=True, waves=8, out_cap=701)


def test_lockstep_decoder_with_the_tail_wavefront():
    """ZPAQ_AMD_TEAM_TAIL=1 (spec_team_kernel.h, round 6: ONE tail wavefront for the scalar work of a workgroup's 8 blocks, the
    mixer wavefronts keep the MIX dot products; measured slower on the MI355X and therefore off by default) must stay
    bit-exact: the lockstep tests above again in a process that generates that form (448 threads per workgroup for -m5)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ZPAQ_AMD_TEAM_TAIL="1")
    chk = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests/emu'); import re, emu, zpaq_amd as z; from zpaq_amd import corpus; "
           "h = z.method_to_header(z.expand_method('5', corpus.block('text', 1 << 20, corpus.BASE_SEED)))[0]; "
           "assert re.findall(r'__launch_bounds__\\((\\d+)\\)', emu.team_source(h)) == ['448']" % (root, root))
    r = subprocess.run([sys.executable, "-c", chk], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-800:]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k", "test_lockstep_decoder and not tail_wavefront"],
                       env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=3000)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-1500:]


# ---------------------------------------------------------------------------------------------------------
# The pipelined encoder (zpaq_amd/csrc/device/pipe_kernel.h): the same generated source the GPU runs, executed
# step by step on the host with consumers launched BEFORE producers inside a step (tests/emu/pipe_emu_main.cpp).

def _pipe_check(oracle, header, inputs, **kw):
    res = emu.pipe_run(header, inputs, **kw)
    for i, (inp, (coded, status, consumed)) in enumerate(zip(inputs, res)):
        assert status == 0 and consumed == len(inp), (i, status)
        assert coded == oracle.encode(header, inp), i


def test_pipe_encoder_standard_chains(zlib_, oracle):
    """-m4 / -m5 chains, ragged blocks incl. empty and one-byte inputs, several chunks per block, more blocks than
    one group, every workgroup width the generator supports -- in both shapes of the encoder (mode 0: a lane per block,
    what a batch that fills the GPU runs; mode 1: MIX / CM / MIX2 with a lane per bit position, what small batches run)."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    h5, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    h4, _, _ = zlib_.method_to_header(zlib_.expand_method("4", blk))
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65])]
    src0, src1 = emu.pipe_source(h5, 64, mode=0), emu.pipe_source(h5, 64, mode=1)
    assert "PIPE_MODE = 0, MIX_BITS = 0" in src0 and "NLIGHT = 9" in src0          # CM, MATCH, MIX2, 4 x SSE, MIX2, coder
    assert "PIPE_MODE = 1, MIX_BITS = 1" in src1 and "NLIGHT = 15" in src1         # CM, MIX2 (20) and SSE as 4 workgroups of 64 lanes each
    for mode in (0, 1):
        _pipe_check(oracle, h5, ragged + [b""], chunk=64, mode=mode)
        _pipe_check(oracle, h5, ragged[:4], chunk=128, group=64, mode=mode)
        _pipe_check(oracle, h5, ragged[:5], chunk=64, group=16, mode=mode)
        _pipe_check(oracle, h4, [b"\0" + corpus.block(kinds[i % 3], 20 + (i * 37) % 180, i).tobytes() for i in range(40)], chunk=64, group=16, mode=mode)
    # the third variant the engine uses: latency shape with its own 2048-byte steps (blocks of 128 KiB and more in
    # production; here blocks that end inside the first, the second and the fifth step)
    assert "PIPE_C = 2048" in emu.pipe_source(h5, None, mode=2) and "PIPE_MODE = 1" in emu.pipe_source(h5, None, mode=2)
    long_ones = [b"\0" + corpus.block(k, n, 5 + i).tobytes() for i, (k, n) in enumerate([("text", 9000), ("records", 5000), ("lcg", 2049), ("zeros", 4096), ("text", 1)])]
    _pipe_check(oracle, h5, long_ones, chunk=None, mode=2)


def test_pipe_encoder_every_component_type_and_legacy_models(oracle, golden):
    seen = set()
    for e in [golden["config_cases"][0]] + golden["level_cases"] + golden["vm_cases"][:4]:
        header = bytes.fromhex(e["header"])
        if header in seen or not header[6] or header[6] > 64:
            continue
        seen.add(header)
        d = gen_input(e).tobytes()
        if len(d) < 64:
            d = corpus.block("records", 600, 3).tobytes()
        for mode in (0, 1):        # (legacy models: MIX lane groups of 2 and 4, several blocks per bit-lane wavefront)
            _pipe_check(oracle, header, [b"\0" + d[:500], b"", d[100:230], b"\0"], chunk=64, mode=mode)
    assert len(seen) >= 6


PIPE_STRESS_CFG = """
comp 3 8 0 0 9
  0 icm 1
  1 isse 2 0
  2 cm 9 255
  3 cm 10 8
  4 match 8 10
  5 mix2 8 0 1 24 255
  6 mix 8 0 6 24 255
  7 sse 8 6 32 255
  8 mix2 9 7 6 16 255
hcomp
  c++ *c=a b=c
  d= 0 *d=a
  d++ a=*b a>>= 1 *d=a
  d++ a=*b a<<= 1 *d=a
  d++ b-- a=*b a+=*c *d=a
  d++ hash *d=a
  d++ a=*c a>>= 6 *d=a
  d++ a=*c a>>= 2 *d=a
  d++ a=*c a>>= 3 *d=a
  d++ a=*c a<<= 1 *d=a
  halt
end
"""


def test_pipe_encoder_table_fetches_that_alias(zlib_, oracle):
    """The units fetch the next bytes' table words before they store this byte's.  Tiny tables and contexts that
    move by little from byte to byte take every aliasing path: same context (forwarding by bit position / from the lane's
    own history), contexts closer than the address range of a byte (fetch after the stores: the bit-lane units' lanes meet
    in memory there), hash rows sharing a 64-byte line.  Both shapes of the encoder."""
    header, _ = zlib_.assemble(PIPE_STRESS_CFG)
    r = np.random.default_rng(1)
    walk = (np.cumsum(r.integers(-3, 4, 700)) & 255).astype(np.uint8).tobytes()
    datas = [walk, corpus.block("text", 600, 5).tobytes(), bytes(500), bytes([7, 7, 8, 8] * 150),
             corpus.block("lcg", 300, 9).tobytes(), bytes(range(256)) * 2]
    # MATCH: long matches, a history buffer (1 KiB here) that wraps, candidates overlapping the byte being written
    rep = bytes(np.random.default_rng(5).integers(0, 256, 97, dtype=np.uint8)) * 40
    more = [rep, corpus.block("text", 1500, 3).tobytes() * 3, bytes(3000), b"abcabcabd" * 400, corpus.block("records", 4000, 8).tobytes()]
    # ... and blocks that FIT the history buffer (MATCH then reads its history from the input: pipe_match_in): long and
    # short repeats, runs of one byte from the block's start on (the reference compares with its zero-filled buffer there),
    # matches that reach back to the first bytes, a match running across chunk boundaries
    fit = [rep[:1000], bytes(1000), b"abcabcabd" * 100, corpus.block("text", 1000, 3).tobytes(), bytes([0, 0, 0, 5]) * 200,
           corpus.block("text", 300, 9).tobytes() * 3, b"\1" * 700, bytes(range(40)) * 20]
    for mode in (0, 1):
        _pipe_check(oracle, header, [b"\0" + d for d in datas], chunk=64, mode=mode)
        _pipe_check(oracle, header, [b"\0" + d for d in more], chunk=256, mode=mode)
        _pipe_check(oracle, header, [b"\0" + d for d in fit], chunk=64, mode=mode)
    _pipe_check(oracle, header, [b"\0" + d for d in fit] + [d[:1023] for d in fit], chunk=128, mode=0, persist=True)


def test_coder_normalisation_in_closed_form(tmp_path):
    """tests/cpp/coder_norm_check.c: the reference's shift-out loop against the closed form of the CODER unit, 2 x 10^7 states."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "coder_norm_check")
    r = subprocess.run(["gcc", "-O2", os.path.join(root, "tests", "cpp", "coder_norm_check.c"), "-o", exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-500:]


MATCH_RING_CFG = """
comp 2 0 0 0 1
  0 match 10 9
hcomp
  b=a a=*d a<<= 8 a+=b *d=a halt
end
"""


def test_match_unit_near_the_end_of_its_history_ring(zlib_, oracle):
    """update0 compares backwards from a candidate with indices modulo the buffer size (libzpaq.cpp:1995-1998): behind a
    candidate near the block's start that is the END of the ring, zero only while nothing has been written there.  A block
    within 255 bytes of the buffer size must therefore go through the buffer in the arena, not through the input
    (pipe_match_any; the advisor's case of round 5: 508 bytes against a 512-byte buffer, a zero run near the end, the bytes
    that follow the block's first bytes repeated behind it).  Lengths around both thresholds, every launch form."""
    header, _ = zlib_.assemble(MATCH_RING_CFG)
    Y, X = 0x59, 0x58

    def case(n):
        d = bytearray(np.random.default_rng(n).integers(1, 256, n, dtype=np.uint8).tobytes())
        d[1:3] = bytes([Y, X])
        d[n - 24] = X
        d[n - 23:n - 10] = bytes(13)
        d[n - 10:n - 8] = bytes([Y, X])
        return bytes(d)

    datas = [case(n) for n in (508, 512, 300, 258, 257, 256, 511, 400)]
    for mode in (0, 1):
        _pipe_check(oracle, header, datas, chunk=64, mode=mode)
    _pipe_check(oracle, header, datas, chunk=64, mode=0, persist=True)


def test_persistent_launch_of_the_pipelined_encoder(zlib_, oracle, golden):
    """The same units inside ONE launch (device/pipe_persist.h): every workgroup of the grid alive at once in the emulator,
    the units waiting for each other through their progress counters, streams stored through the write-through accessors,
    CM / MIX2 tables and the ICM / ISSE side tables resident in LDS from chunk to chunk.  Standard chains in both shapes
    (several groups, ragged blocks, empty blocks, more chunks than ring slots), the aliasing stress chain, every component
    type and the legacy models."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    h5, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    h4, _, _ = zlib_.method_to_header(zlib_.expand_method("4", blk))
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65])]
    assert "zpq_pipe_persist" in emu.pipe_source(h5, 64, mode=0) and "PS_WPG" in emu.pipe_source(h5, 64, mode=1)
    for mode in (0, 1):
        _pipe_check(oracle, h5, ragged + [b""], chunk=64, mode=mode, persist=True)
        _pipe_check(oracle, h4, [b"\0" + corpus.block(kinds[i % 3], 20 + (i * 37) % 180, i).tobytes() for i in range(40)], chunk=64, group=16, mode=mode, persist=True)
    # variant 3 (round 6): the latency shape with a wavefront per SIMD -- workgroups of 4, twice as many per group
    assert "PS_WAVES = 4, PS_WPG = 28" in emu.pipe_source(h5, 64, mode=3) and "PS_WAVES = 8, PS_WPG = 14" in emu.pipe_source(h5, 64, mode=1)
    _pipe_check(oracle, h5, ragged + [b""], chunk=64, mode=3, persist=True)
    # more chunks than ring slots: producers have to wait for their consumers' progress
    _pipe_check(oracle, h5, [b"\0" + corpus.block("text", 64 * 45 + 7, 3).tobytes(), b"\0" + corpus.block("records", 64 * 30, 4).tobytes()], chunk=64, mode=0, persist=True)
    header, _ = zlib_.assemble(PIPE_STRESS_CFG)
    r = np.random.default_rng(1)
    walk = (np.cumsum(r.integers(-3, 4, 700)) & 255).astype(np.uint8).tobytes()
    rep = bytes(np.random.default_rng(5).integers(0, 256, 97, dtype=np.uint8)) * 40
    datas = [walk, corpus.block("text", 600, 5).tobytes(), bytes(500), bytes([7, 7, 8, 8] * 150), rep, b"abcabcabd" * 400]
    for mode in (0, 1):
        _pipe_check(oracle, header, [b"\0" + d for d in datas], chunk=64, mode=mode, persist=True)
    seen = set()
    for e in [golden["config_cases"][0]] + golden["level_cases"]:
        hdr = bytes.fromhex(e["header"])
        if hdr in seen or not hdr[6] or hdr[6] > 64:
            continue
        seen.add(hdr)
        d = gen_input(e).tobytes()
        if len(d) < 64:
            d = corpus.block("records", 600, 3).tobytes()
        _pipe_check(oracle, hdr, [b"\0" + d[:500], b"", d[100:230], b"\0"], chunk=64, mode=0, persist=True)
    assert len(seen) >= 3


SMALL_CHAIN_CFGS = [
    # (config, blocks).  One ICM; ICM + ISSE on tables of 4 KiB (the ROW units stay lane-per-block: the nibbles of a byte can share
    # a line) and of 64 KiB (a lane per nibble); two ISSEs in a row behind a CM and a MATCH.
    "comp 1 0 0 0 1\n  0 icm 12\nhcomp\n  *d=a halt\nend\n",
    "comp 2 0 0 0 2\n  0 icm 4\n  1 isse 4 0\nhcomp\n  b=a a=*d a<<= 4 a+=b *d=a d++ a<<= 3 a+=b *d=a halt\nend\n",
    "comp 2 0 0 0 2\n  0 icm 10\n  1 isse 10 0\nhcomp\n  b=a a=*d a<<= 8 a+=b *d=a d++ a<<= 5 a+=b *d=a halt\nend\n",
    "comp 2 0 0 0 4\n  0 cm 9 255\n  1 icm 9\n  2 isse 10 1\n  3 isse 11 2\nhcomp\n  b=a *d=a d++ a=*d a<<= 8 a+=b *d=a d++ a<<= 2 a+=b *d=a d++ hash *d=a halt\nend\n",
    # an M array of 8 bytes, written and read back by the program (HCOMP keeps it in LDS inside the persistent launch)
    "comp 2 3 0 0 2\n  0 icm 10\n  1 isse 10 0\nhcomp\n  c++ *c=a b=c a=0 d=0 hash b-- hash *d=a d++ b-- hash *d=a halt\nend\n",
]


def test_small_chains_in_the_latency_shape(zlib_, oracle):
    """Round 6: the latency shape of a chain of at most 16 unit wavefronts (configs[1]'s n = 2 is six) -- a wavefront per SIMD
    (workgroups of 4), ISSE pairs as two words, whole squash / stretch tables in LDS, ROW units with a lane per nibble on tables of
    8 KiB and more -- and the coder every latency-shape launch now has (pipe_coder_fast: one 4-byte store per bit, a byte coded
    again with the reference's loop when its test fails or when less than 40 bytes of room are left).  Ragged and empty blocks,
    zeros (every next row clashes with the row being stored), incompressible bytes (the coder emits at nearly every bit), blocks
    against their output capacity (status 3 exactly when the reference's length does not fit, never a byte past it)."""
    blk = corpus.block("lcg", 1 << 18, corpus.BASE_SEED)
    h3, _, _ = zlib_.method_to_header(zlib_.expand_method("3", blk))
    src = emu.pipe_source(h3, 64, mode=1)
    assert "PS_CODER_FAST = true, PS_SMALL = true" in src and "PS_WAVES = 4" in src
    assert "PS_SMALL = false" in emu.pipe_source(h3, 64, mode=0) and "PS_CODER_FAST = false" in emu.pipe_source(h3, 64, mode=0)
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65, 2000, 777])]
    _pipe_check(oracle, h3, ragged + [b""], chunk=64, mode=1, persist=True)
    _pipe_check(oracle, h3, [b"\0" + corpus.block(kinds[i % 5], 100 + 13 * i, i).tobytes() for i in range(70)], chunk=128, mode=1, persist=True)     # three groups
    for cfg in SMALL_CHAIN_CFGS:
        header, _ = zlib_.assemble(cfg)
        assert "PS_SMALL = true" in emu.pipe_source(header, 64, mode=1)
        _pipe_check(oracle, header, ragged, chunk=64, mode=1, persist=True)
    # long incompressible blocks: ~2 x 10^6 coded bits, a few dozen of which fail the fast form's test
    long_ones = [corpus.block("lcg", 8192, 100 + i).tobytes() for i in range(32)]
    _pipe_check(oracle, h3, long_ones, chunk=512, mode=1, persist=True)
    # output capacities around the coded length: the careful path near the end, status 3 when it does not fit
    data = [b"\0" + corpus.block("lcg", 400, 7).tobytes(), b"\0" + corpus.block("text", 400, 8).tobytes(), b"\0" + bytes(300)]
    want = [oracle.encode(h3, d) for d in data]
    for cap in (8, 40, 41, len(want[1]), len(want[0]) - 1, len(want[0]), len(want[0]) + 3, len(want[0]) + 39, len(want[0]) + 41):
        res = emu.pipe_run(h3, data, chunk=64, mode=1, persist=True, out_cap=cap)
        for w, (coded, status, consumed) in zip(want, res):
            assert status == (0 if len(w) <= cap else 3), (cap, len(w), status)
            assert coded == w[:cap]


def test_pipe_units_do_not_depend_on_lane_order(zlib_, oracle, monkeypatch):
    """Between two cross-lane operations the emulator may run the lanes of a wavefront in any order; the hardware runs them
    together.  The bit-lane units' lanes meet in memory (the positions of one block share its tables), so they must give
    the oracle's bytes whatever the order: the default one, reversed, and two shuffles."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    h5, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    header, _ = zlib_.assemble(PIPE_STRESS_CFG)
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65])]
    r = np.random.default_rng(1)
    walk = (np.cumsum(r.integers(-3, 4, 700)) & 255).astype(np.uint8).tobytes()
    stress = [b"\0" + d for d in (walk, bytes(500), bytes([7, 7, 8, 8] * 150), bytes(range(256)) * 2)]
    for order in ("reverse", "shuffle:3", "shuffle:4"):
        monkeypatch.setenv("ZPQ_EMU_ORDER", order)
        for mode in (0, 1):
            _pipe_check(oracle, h5, ragged, chunk=64, mode=mode)
            _pipe_check(oracle, header, stress, chunk=64, mode=mode)


def test_random_models_through_both_coders(zlib_, oracle):
    """tests/fuzz_emu.py, two models: random components of every type with random parameters and a random HCOMP program,
    through the per-header wavefront coder (both ways), the decoder with two blocks per wavefront and the pipelined encoder
    (both shapes), against the oracle.
    (240 models of seeds 2 and 3 were run when this was added: no mismatch.)"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import fuzz_emu
    assert fuzz_emu.run(2, 20260926, verbose=False) == 0


def _host_tokens(L, xm, data):
    import ctypes as C
    u8p = C.POINTER(C.c_ubyte)
    L.zpq_lz77_tokens_host.argtypes = [C.c_char_p, u8p, C.c_uint32, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t)]
    buf = np.frombuffer(bytearray(data), np.uint8).copy() if data else np.zeros(1, np.uint8)
    cap = len(data) + 4
    toks = np.zeros(4 * cap, np.uint32)
    cnt = C.c_size_t(0)
    assert L.zpq_lz77_tokens_host(xm.encode(), buf.ctypes.data_as(u8p), len(data), toks.ctypes.data_as(C.POINTER(C.c_uint32)), cap, C.byref(cnt)) == 0, L.zpq_last_error()
    return toks[:4 * cnt.value].tobytes(), buf[:len(data)].tobytes()


def test_lz77_parse_and_bwt_kernels_against_the_host(zlib_):
    """device/lz77_kernel.h on the emulator: the search (one lane per position, both values of the pending-literals bit), the
    walk (one wavefront per block, v_readlane chain) and the BWT emit must give the host's list of matches / BWT stream -- whose
    coded form the reference's archives pin (test_host.py) -- for both code levels, look-aheads 0..3, small and large buckets,
    E8E9, ragged and tiny inputs, long repeats, and a block that crosses the inverse array's 2^17-position window."""
    import ctypes as C
    L = zlib_.lib()
    u8p = C.POINTER(C.c_ubyte)
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ins = [corpus.block(kinds[i % 5], n, 100 + i).tobytes() if n else b"" for i, n in enumerate([3000, 1, 2, 70, 257, 5000, 9000, 0, 4097])]
    ins.append(b"abcdefghij" * 700 + corpus.block("lcg", 500, 3).tobytes() + b"abcdefghij" * 300)
    for xm in ("x0,2,5,0,7,21,1c0,0,511", "x0,1,4,0,3,21,1", "x0,2,12,0,7,21,1c0,0,511i2", "x0,6,5,0,2,21,0c0,0,511", "x0,2,4,0,7,21,3c0,0,511",
               "x0,5,5,0,5,21,2"):
        a = zlib_.method_to_header(xm)[2]
        host = [_host_tokens(L, xm, d) for d in ins]
        got = emu.lz77_run(a[1] & 3, a[2], a[6], (1 << a[4]) - 1, 17 + a[0], [h[1] for h in host])
        for k, (h, g) in enumerate(zip(host, got)):
            assert h[0] == g, (xm, k, len(ins[k]), len(h[0]) // 16, len(g) // 16)
    big = corpus.block("text", 140000, 77).tobytes()
    for xm in ("x0,2,5,0,7,21,1c0,0,511", "x0,1,4,0,3,21,2"):
        a = zlib_.method_to_header(xm)[2]
        ht, buf = _host_tokens(L, xm, big)
        assert emu.lz77_run(a[1] & 3, a[2], a[6], (1 << a[4]) - 1, 17 + a[0], [buf])[0] == ht
    # the BWT stream preprocess_block makes
    L.zpq_preprocess_block.argtypes = [C.c_char_p, u8p, C.c_uint32, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    bw = [corpus.block(kinds[i % 5], n, 300 + i).tobytes() for i, n in enumerate([1, 2, 3, 1000, 4096, 7777])]
    for d, g in zip(bw, emu.lz77_run(3, 0, 0, 0, 0, bw)):
        buf = np.frombuffer(bytearray(d), np.uint8).copy()
        out = np.zeros(len(d) + 16, np.uint8)
        ln = C.c_size_t(0)
        assert L.zpq_preprocess_block(b"x0,3ci1", buf.ctypes.data_as(u8p), len(d), out.ctypes.data_as(u8p), out.size, C.byref(ln)) == 0
        assert out[:ln.value].tobytes() == g, (len(d), ln.value, len(g))
