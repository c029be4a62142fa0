#!/usr/bin/env python3
"""Timeline of the pipelined encoder out of a rocprofv3 rocpd database (--kernel-trace): the step structure.

  python profiles/pipe_timeline.py gpurun_out/prof/x_results.db [--split]

Groups the zpq_pipe_* dispatches into steps (one launch of each kernel per step; with ZPAQ_AMD_PIPE_SPLIT=1 one launch
per UNIT, told apart by their order inside the step), and prints, per kernel / unit, the average and maximum duration,
how often it was the last one to finish in its step, the average step span and the average gap between steps."""
import collections
import sqlite3
import sys


def main(db_path, split):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, start, end, grid_x, workgroup_x from kernels where name like 'zpq_pipe_%' order by start").fetchall()
    if not rows:
        print("no zpq_pipe_* dispatches in", db_path)
        return
    # a step starts with its hcomp launch... not necessarily first in time; cluster by gaps instead: all launches of a step
    # overlap, steps are separated by the join
    steps, cur_step, cur_end = [], [], None
    for r in rows:
        if cur_end is not None and r[1] >= cur_end - 500:       # starts after (almost) everything before ended: next step
            steps.append(cur_step)
            cur_step, cur_end = [], None
        cur_step.append(r)
        cur_end = r[2] if cur_end is None else max(cur_end, r[2])
    steps.append(cur_step)
    full = [s for s in steps if len(s) == max(len(x) for x in steps)]
    print(f"{len(rows)} dispatches, {len(steps)} steps ({len(full)} with all units busy)")
    dur = collections.defaultdict(list)
    last = collections.Counter()
    spans, gaps = [], []
    prev_end = None
    for s in full:
        seen = collections.Counter()
        ends = []
        for name, st, en, gx, wx in s:
            key = name.replace("zpq_pipe_", "")
            if split:
                key += f"[{seen[name]}]"
                seen[name] += 1
            dur[key].append((en - st) / 1e3)
            ends.append((en, key))
        last[max(ends)[1]] += 1
        s0, s1 = min(x[1] for x in s), max(x[2] for x in s)
        spans.append((s1 - s0) / 1e3)
        if prev_end is not None:
            gaps.append((s0 - prev_end) / 1e3)
        prev_end = s1
    print(f"step span: avg {sum(spans) / len(spans):.1f} us, gap between steps: avg {sum(gaps) / max(len(gaps), 1):.1f} us")
    print(f"{'unit':<14}{'avg us':>10}{'max us':>10}{'last in step':>14}")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
        print(f"{k:<14}{sum(v) / len(v):>10.1f}{max(v):>10.1f}{last[k]:>14}")


if __name__ == "__main__":
    main(sys.argv[1], "--split" in sys.argv)
