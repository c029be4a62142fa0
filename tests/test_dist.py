"""CPU, world_size 2 and 3, gloo: the N > 1 path of the framework -- block sharding, scatter of the
corpus, gather of variable-length archives in block order, max-over-ranks timing.  Every rank runs the PRODUCT's
compress_blocks on its shard with method "2" (LZ77 without a context model: that path is host-only, so it works
without a GPU) and rank 0 checks the gathered archives, in block order, against a single-process run, against a round
trip and -- when oracle/_ref is built -- against the reference's compressBlock.  What runs between the ranks is exactly
what runs on the 8-GPU node; there the method is "5" and the coder the device."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT


def test_shard_range_covers_everything():
    from zpaq_amd.dist import shard_range
    for nb in (0, 1, 7, 8, 1024, 8191):
        for w in (1, 2, 3, 8):
            spans = [shard_range(nb, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == nb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


WORKER = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, os.environ["ZROOT"])
    import zpaq_amd as z
    from zpaq_amd import corpus, dist as zd
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    NB, BS = 7, 30000                      # odd count: ranks get different numbers of blocks
    blocks = corpus.corpus("text", NB, BS) if rank == 0 else None
    mine = zd.scatter_blocks(blocks, NB, BS)
    b, e = zd.shard_range(NB, rank, world)
    assert mine.shape == (e - b, BS)
    ref = corpus.corpus("text", NB, BS)[b:e]
    assert (mine.numpy() == ref).all(), "scatter delivered the wrong slice"
    archives = z.compress_blocks([row for row in mine.numpy()], "2")
    zd.barrier()
    t = zd.max_over_ranks(0.5 + rank)
    assert abs(t - (world - 0.5)) < 1e-9
    got = zd.gather_archives(archives)
    if rank == 0:
        assert len(got) == NB
        full = corpus.corpus("text", NB, BS)
        alone = z.compress_blocks([row for row in full], "2")
        for k in range(NB):
            assert got[k] == alone[k], "archives out of block order"
            assert z.decompress(got[k], BS + 16) == full[k].tobytes()
        assert len(set(got)) == NB and max(len(a) for a in got) < BS          # distinct blocks, really compressed
        sys.path.insert(0, os.path.join(os.environ["ZROOT"]))
        from oracle.oracle_py import Ref, have_ref
        if have_ref():
            r = Ref()
            for k in range(NB):
                assert got[k] == r.compress_block(full[k].tobytes(), "2"), "not the reference's archive"
            print("REFERENCE_OK")
        print("RANK0_OK")
    else:
        assert got is None
    dist.destroy_process_group()
""")


@pytest.mark.parametrize("world", [2, 3])
def test_scatter_gather_gloo(tmp_path, world):
    """world_size 2 and 3 over 7 blocks: 4 + 3 and 3 + 2 + 2 -- uneven splits both ways."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), ZROOT=ROOT, GLOO_SOCKET_IFNAME="lo")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\\n".join(outs)
    assert "RANK0_OK" in outs[0]
    from oracle.oracle_py import have_ref
    if have_ref():
        assert "REFERENCE_OK" in outs[0]


def test_torch_corpus_generator_on_cpu():
    import torch
    from zpaq_amd import corpus, corpus_torch
    o = corpus_torch.text_blocks(3, 50000, 777, torch.device("cpu"), chunk=2).numpy()
    for b in range(3):
        assert (o[b] == corpus.zipf_text(50000, 777 + b)).all()


def test_bench_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` starts its own N ranks (torch.distributed.run, RCCL); with fewer than N GPUs visible it
    must say so instead of running everything on one device and printing n_gpus: 1 (here: no GPU at all)."""
    import torch
    if torch.cuda.device_count() >= 2:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "--gpus 2 requested but only" in (r.stderr + r.stdout), (r.stdout + r.stderr)[-800:]
    # ... and under a launcher whose world size disagrees with --gpus
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env2)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_dry_run_is_the_whole_rank_flow_on_cpu():
    """`bench.py --gpus 2 --dry-run`: what a multi-GPU run adds around the hot path, on CPU (gloo, two ranks, a method
    without a model): the corpus made on rank 0 and scattered in uneven contiguous ranges, every rank coding its range with
    no collective, the timed region between barriers with the maximum over the ranks, the archives gathered in block
    order and decoded back on rank 0, ONE JSON line that carries n_gpus and dist_ms."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0, (r.stdout + r.stderr)[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] and d["all_status_ok"] and d["archives_in_block_order_and_round_trip"]
    assert d["config"]["blocks_total"] == 25 and d["dist_ms"]["scatter_bytes"] == 25 * 65536 and d["dist_ms"]["gather_bytes"] > 0
