#!/usr/bin/env python3
"""Timeline of the pipelined encoder out of a rocprofv3 rocpd database (--kernel-trace).

  python profiles/pipe_timeline.py gpurun_out/prof/x_results.db

The encoder launches six kernels per step on six in-order streams that run up to `slack` steps apart, so there is no
global step boundary to cut at.  Per coding sequence (dispatches of zpq_pipe_* separated from the next sequence by an
init_arena_kernel) this prints: the span (first start -> last end: what bench.py's hipEvents measure as `kernel_ms.code`),
and per kernel the number of launches, the average / maximum duration and the busy time (sum of durations) -- the
kernel whose busy time comes closest to the span is the one that paces the run."""
import collections
import sqlite3
import sys


def main(db_path):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, start, end from kernels where name like 'zpq_pipe_%' or name like '%init_arena%' order by start").fetchall()
    seqs, cur_seq = [], []
    for name, st, en in rows:
        if "init_arena" in name:
            if cur_seq:
                seqs.append(cur_seq)
            cur_seq = []
        else:
            cur_seq.append((name.replace("zpq_pipe_", ""), st, en))
    if cur_seq:
        seqs.append(cur_seq)
    for i, s in enumerate(seqs):
        span = (max(x[2] for x in s) - min(x[1] for x in s)) / 1e6
        print(f"sequence {i}: {len(s)} launches, span {span:.1f} ms")
        by = collections.defaultdict(list)
        for name, st, en in s:
            by[name].append((en - st) / 1e3)
        print(f"  {'kernel':<8}{'launches':>9}{'avg us':>10}{'max us':>10}{'busy ms':>10}{'busy/span':>10}")
        for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            print(f"  {k:<8}{len(v):>9}{sum(v) / len(v):>10.1f}{max(v):>10.1f}{sum(v) / 1e3:>10.1f}{sum(v) / 1e3 / span:>10.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
