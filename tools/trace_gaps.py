#!/usr/bin/env python3
"""GPU busy fraction and idle gaps of the timed steps of a `rocprofv3 --kernel-trace` run of bench.py.

    python3 tools/trace_gaps.py <dir with *_kernel_trace.csv> [--steps K] [--marker KERNEL_SUBSTRING]

The timed region of bench.py is its last K optimiser steps before the roofline leg.  A step is
delimited by a marker kernel that runs exactly once per step (default: the proposal decode kernel of
the hot path); the K intervals between the last K+1 markers *before the leg's first launch* are the
steps.  Per step: span, sum of kernel durations on the busiest queue overlap-merged across queues
(busy), idle = span - busy, and the largest gaps with the kernels on either side.  Also asserts what
VERDICT r2 asked for: no MIOpen `naive_conv_*` kernel in the trace (their presence means the find
search ran under the profiler and the trace is not the step the bench times).
"""
import argparse
import csv
import glob
import json
import os


def load(root):
    rows = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def merged_busy(rows):
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--marker", default="proposal_decode")
    ap.add_argument("--leg-marker", default="roi_windows_kernel")
    ap.add_argument("--allow-naive-conv", action="store_true")
    args = ap.parse_args()
    rows = load(args.root)
    assert rows, "no kernel trace under %s" % args.root
    naive = [n for _, _, n in rows if "naive_conv" in n]
    marks = [i for i, r in enumerate(rows) if args.marker in r[2]]
    assert len(marks) > args.steps, "marker %r found %d times" % (args.marker, len(marks))
    # the roofline leg launches the pair outside any step: its markers come after the last step's decode
    marks = marks[-(args.steps + 1):] if len(marks) == args.steps + 1 else marks
    # keep the last K+1 markers that are followed by another marker within a plausible step
    steps = []
    for a, b in zip(marks[:-1], marks[1:]):
        steps.append(rows[a:b])
    steps = steps[-args.steps:]
    out = []
    for st in steps:
        span = st[-1][1] - st[0][0]
        # the step's span = first kernel start .. next step's first kernel start
        busy = merged_busy(st)
        gaps = []
        end = st[0][1]
        prev = st[0][2]
        for s, e, n in st[1:]:
            if s > end:
                gaps.append((s - end, prev[:60], n[:60]))
            if e > end:
                end, prev = e, n
        gaps.sort(reverse=True)
        out.append(dict(span_ms=round(span / 1e6, 3), busy_ms=round(busy / 1e6, 3), idle_ms=round((span - busy) / 1e6, 3),
                        busy_frac=round(busy / span, 4), kernels=len(st),
                        top_gaps=[dict(us=round(g / 1e3, 1), after=a, before=b) for g, a, b in gaps[:5]]))
    res = dict(steps=out, naive_conv_kernels=len(naive),
               mean_busy_frac=round(sum(o["busy_frac"] for o in out) / len(out), 4),
               mean_idle_ms=round(sum(o["idle_ms"] for o in out) / len(out), 3),
               mean_span_ms=round(sum(o["span_ms"] for o in out) / len(out), 3))
    print(json.dumps(res, indent=1))
    if naive and not args.allow_naive_conv:
        raise SystemExit("naive_conv_* kernels in the trace: the MIOpen find search ran under the profiler")


if __name__ == "__main__":
    main()
