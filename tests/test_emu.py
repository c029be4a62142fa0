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
# The pipelined encoder (zpaq_amd/csrc/device/pipe_kernel.h): the same generated source the GPU runs, executed
# step by step on the host with consumers launched BEFORE producers inside a step (tests/emu/pipe_emu_main.cpp).

def _pipe_check(oracle, header, inputs, **kw):
    res = emu.pipe_run(header, inputs, **kw)
    for i, (inp, (coded, status, consumed)) in enumerate(zip(inputs, res)):
        assert status == 0 and consumed == len(inp), (i, status)
        assert coded == oracle.encode(header, inp), i


def test_pipe_encoder_standard_chains(zlib_, oracle):
    """-m4 / -m5 chains, ragged blocks incl. empty and one-byte inputs, several chunks per block, more blocks than
    one group, and every workgroup width the generator supports."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    h5, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    h4, _, _ = zlib_.method_to_header(zlib_.expand_method("4", blk))
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65])]
    _pipe_check(oracle, h5, ragged + [b""], chunk=64)
    _pipe_check(oracle, h5, ragged[:4], chunk=128, group=64)
    _pipe_check(oracle, h4, [b"\0" + corpus.block(kinds[i % 3], 20 + (i * 37) % 180, i).tobytes() for i in range(40)], chunk=64, group=16)


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
        _pipe_check(oracle, header, [b"\0" + d[:500], b"", d[100:230], b"\0"], chunk=64)
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
    """The units fetch the next byte's table words before they store this byte's.  Tiny tables and contexts that
    move by little from byte to byte take every aliasing path: same context (forwarding by bit position), contexts
    closer than the address range of a byte (fetch after the stores), hash rows sharing a 64-byte line."""
    header, _ = zlib_.assemble(PIPE_STRESS_CFG)
    r = np.random.default_rng(1)
    walk = (np.cumsum(r.integers(-3, 4, 700)) & 255).astype(np.uint8).tobytes()
    datas = [walk, corpus.block("text", 600, 5).tobytes(), bytes(500), bytes([7, 7, 8, 8] * 150),
             corpus.block("lcg", 300, 9).tobytes(), bytes(range(256)) * 2]
    _pipe_check(oracle, header, [b"\0" + d for d in datas], chunk=64)
    # MATCH: long matches, a history buffer (1 KiB here) that wraps, candidates overlapping the byte being written
    rep = bytes(np.random.default_rng(5).integers(0, 256, 97, dtype=np.uint8)) * 40
    more = [rep, corpus.block("text", 1500, 3).tobytes() * 3, bytes(3000), b"abcabcabd" * 400, corpus.block("records", 4000, 8).tobytes()]
    _pipe_check(oracle, header, [b"\0" + d for d in more], chunk=256)


def test_pipe_encoder_mix_bit_lanes(zlib_, oracle, golden):
    """ZPAQ_AMD_PIPE_MIX_BITS=1: the MIX unit with a lane per (block, bit position, weight quad) and rows fetched
    MIX_DEPTH bytes ahead (pipe_kernel.h::pipe_mix_bits_body).  Same bytes as the oracle for the -m5 chain at every
    depth, for the legacy models (lane groups of 2 and 4: several blocks per wavefront), for other group widths, and
    for the stress chain whose 256-row MIX makes every pair of different contexts collide (the fetch-again path) while
    equal contexts exercise the lane's own history."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    h5, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65])]
    for depth in (1, 4):
        assert "MIX_BITS = 1, MIX_DEPTH = %d" % depth in emu.pipe_source(h5, 64, None, None, 1, depth)
        _pipe_check(oracle, h5, ragged + [b""], chunk=64, mix_bits=1, mix_depth=depth)
    _pipe_check(oracle, h5, ragged[:4], chunk=128, group=64, mix_bits=1)
    _pipe_check(oracle, h5, ragged[:5], chunk=64, group=16, mix_bits=1, mix_depth=2)
    # legacy mid / max models and the nine-type config
    seen = set()
    for e in [golden["config_cases"][0]] + golden["level_cases"]:
        header = bytes.fromhex(e["header"])
        if header in seen or not header[6] or header[6] > 64:
            continue
        seen.add(header)
        if "MIX_BITS = 1" not in emu.pipe_source(header, 64, None, None, 1, 3):
            continue                            # no MIX, or one that does not keep the whole partial byte in its row index
        d = gen_input(e).tobytes()
        if len(d) < 64:
            d = corpus.block("records", 600, 3).tobytes()
        _pipe_check(oracle, header, [b"\0" + d[:500], b"", d[100:230], b"\0"], chunk=64, mix_bits=1)
    header, _ = zlib_.assemble(PIPE_STRESS_CFG)
    r = np.random.default_rng(1)
    walk = (np.cumsum(r.integers(-3, 4, 700)) & 255).astype(np.uint8).tobytes()
    datas = [walk, corpus.block("text", 600, 5).tobytes(), bytes(500), bytes([7, 7, 8, 8] * 150),
             corpus.block("lcg", 300, 9).tobytes(), bytes(range(256)) * 2]
    for depth in (2, 4):
        _pipe_check(oracle, header, [b"\0" + d for d in datas], chunk=64, mix_bits=1, mix_depth=depth)


def test_pipe_encoder_light_bit_lanes(zlib_, oracle, golden):
    """ZPAQ_AMD_PIPE_LIGHT_BITS=7 (1 CM | 2 MIX2 | 4 SSE): CM, MIX2 and SSE with a lane per (block, bit position), workgroups of 8 blocks x 8 positions,
    unit, table words fetched LIGHT_DEPTH bytes ahead (pipe_kernel.h::pipe_cm_bits / pipe_mix2_bits / pipe_sse_bits) --
    alone and together with the bit-lane MIX, on the -m5 chain, the legacy models and the stress chain whose tiny tables
    make words of different contexts collide."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    h5, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65])]
    src = emu.pipe_source(h5, 64, None, None, None, None, 7, 2)
    assert "NLIGHT = 15" in src and "LIGHT_DEPTH = 2" in src           # CM, MIX2 (20) and SSE as 4 workgroups of 64 lanes each
    for depth in (1, 4):
        _pipe_check(oracle, h5, ragged + [b""], chunk=64, light_bits=7, light_depth=depth)
    _pipe_check(oracle, h5, ragged, chunk=64, light_bits=7, light_depth=3, mix_bits=1, mix_depth=3)
    _pipe_check(oracle, h5, ragged[:4], chunk=128, group=64, light_bits=7, mix_bits=1)
    _pipe_check(oracle, h5, ragged[:5], chunk=64, group=16, light_bits=7, light_depth=2)
    seen = set()
    for e in [golden["config_cases"][0]] + golden["level_cases"]:
        header = bytes.fromhex(e["header"])
        if header in seen or not header[6] or header[6] > 64:
            continue
        seen.add(header)
        d = gen_input(e).tobytes()
        if len(d) < 64:
            d = corpus.block("records", 600, 3).tobytes()
        _pipe_check(oracle, header, [b"\0" + d[:500], b"", d[100:230], b"\0"], chunk=64, light_bits=7, mix_bits=1)
    header, _ = zlib_.assemble(PIPE_STRESS_CFG)
    r = np.random.default_rng(1)
    walk = (np.cumsum(r.integers(-3, 4, 700)) & 255).astype(np.uint8).tobytes()
    datas = [walk, corpus.block("text", 600, 5).tobytes(), bytes(500), bytes([7, 7, 8, 8] * 150),
             corpus.block("lcg", 300, 9).tobytes(), bytes(range(256)) * 2]
    for depth in (2, 4):
        _pipe_check(oracle, header, [b"\0" + d for d in datas], chunk=64, light_bits=7, light_depth=depth, mix_bits=1, mix_depth=depth)


def test_pipe_encoder_row_nibble_lanes(zlib_, oracle, golden):
    """ZPAQ_AMD_PIPE_ROW_NIBBLES=1: the ROW units with a lane per (block, nibble), candidate rows fetched ROW_DEPTH bytes
    ahead (pipe_kernel.h::pipe_row_nibbles) -- alone and with every other experimental unit on.  The stress chain's hash
    tables of 2 and 4 lines make the two nibbles of a byte share a line (second pass) and nearly every fetch stale; zeros
    and the repeated patterns keep the context constant (fetch-again path on every byte)."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    h5, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65])]
    assert "ROW_NIBBLES = 1, ROW_DEPTH = 3" in emu.pipe_source(h5, 64, row_nibbles=1, row_depth=3)
    for depth in (1, 4):
        _pipe_check(oracle, h5, ragged + [b""], chunk=64, row_nibbles=1, row_depth=depth)
    everything = dict(row_nibbles=1, mix_bits=1, light_bits=7, full_squash=1)
    _pipe_check(oracle, h5, ragged[:4], chunk=64, full_squash=1)            # ZPAQ_AMD_PIPE_FULL_SQUASH alone: whole squash table in LDS
    _pipe_check(oracle, h5, ragged, chunk=64, row_flat=1)                   # ZPAQ_AMD_PIPE_ROW_FLAT: one-lane ROW unit, row picked by masks
    # ZPAQ_AMD_PIPE_MAP_ILP: two / four blocks per lane in the ICM and ISSE maps (ragged lengths: the blocks of a lane end apart)
    many = [b"\0" + corpus.block(kinds[i % 5], 40 + (i * 37) % 200, i).tobytes() for i in range(40)]
    _pipe_check(oracle, h5, many, chunk=64, map_ilp=2)
    _pipe_check(oracle, h5, ragged + [b""], chunk=64, map_ilp=4)
    _pipe_check(oracle, h5, many[:20], chunk=64, map_ilp=2, **everything)
    _pipe_check(oracle, h5, ragged, chunk=64, **everything)
    _pipe_check(oracle, h5, ragged[:5], chunk=64, group=16, row_nibbles=1, row_depth=2)
    seen = set()
    for e in [golden["config_cases"][0]] + golden["level_cases"]:
        header = bytes.fromhex(e["header"])
        if header in seen or not header[6] or header[6] > 64:
            continue
        seen.add(header)
        d = gen_input(e).tobytes()
        if len(d) < 64:
            d = corpus.block("records", 600, 3).tobytes()
        _pipe_check(oracle, header, [b"\0" + d[:500], b"", d[100:230], b"\0"], chunk=64, **everything)
    header, _ = zlib_.assemble(PIPE_STRESS_CFG)
    r = np.random.default_rng(1)
    walk = (np.cumsum(r.integers(-3, 4, 700)) & 255).astype(np.uint8).tobytes()
    datas = [walk, corpus.block("text", 600, 5).tobytes(), bytes(500), bytes([7, 7, 8, 8] * 150),
             corpus.block("lcg", 300, 9).tobytes(), bytes(range(256)) * 2]
    for depth in (1, 3):
        _pipe_check(oracle, header, [b"\0" + d for d in datas], chunk=64, row_nibbles=1, row_depth=depth)
    _pipe_check(oracle, header, [b"\0" + d for d in datas], chunk=64, **everything)


def test_pipe_units_do_not_depend_on_lane_order(zlib_, oracle, monkeypatch):
    """Between two cross-lane operations the emulator may run the lanes of a wavefront in any order; the hardware runs them
    together.  The experimental units' lanes meet in memory (the positions of one block share its tables), so they must give
    the oracle's bytes whatever the order: the default one, reversed, and two shuffles."""
    blk = corpus.block("text", 1 << 20, corpus.BASE_SEED)
    h5, _, _ = zlib_.method_to_header(zlib_.expand_method("5", blk))
    header, _ = zlib_.assemble(PIPE_STRESS_CFG)
    kinds = ["text", "lcg", "zeros", "records", "pattern"]
    ragged = [b"\0" + corpus.block(kinds[i % 5], n, 40 + i).tobytes() for i, n in enumerate([300, 150, 200, 97, 0, 1, 63, 64, 65])]
    r = np.random.default_rng(1)
    walk = (np.cumsum(r.integers(-3, 4, 700)) & 255).astype(np.uint8).tobytes()
    stress = [b"\0" + d for d in (walk, bytes(500), bytes([7, 7, 8, 8] * 150), bytes(range(256)) * 2)]
    everything = dict(row_nibbles=1, mix_bits=1, light_bits=7, full_squash=1)
    for order in ("reverse", "shuffle:3", "shuffle:4"):
        monkeypatch.setenv("ZPQ_EMU_ORDER", order)
        _pipe_check(oracle, h5, ragged, chunk=64, **everything)
        _pipe_check(oracle, header, stress, chunk=64, **everything)
        _pipe_check(oracle, h5, ragged[:4], chunk=64)                   # and the product's own configuration
