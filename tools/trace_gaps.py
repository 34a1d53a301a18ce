#!/usr/bin/env python3
"""GPU busy fraction and idle gaps of the timed steps of a `rocprofv3 --kernel-trace` run of bench.py.

    python3 tools/trace_gaps.py <dir with *_kernel_trace.csv> [--steps K] [--marker KERNEL_SUBSTRING]

A step is delimited by a marker kernel that runs a fixed number of times per step (default: the
proposal decode kernel of the hot path, once; twice per alternating iteration); the timed steps start
at marker number warmup * markers_per_step.  Per step: span, kernel time overlap-merged across queues (busy), idle = span - busy, and the
largest gaps with the kernels on either side; --stats-csv writes the per-kernel totals of the TIMED
STEPS ONLY.  rocprofv3's own *_kernel_stats.csv covers the whole process, and MIOpen runs its
naive_conv_* reference kernels for every new convolution during bench.py's warm-up steps (seconds of
GPU time, 94 % of a whole-process summary) -- they must not appear inside the timed steps, which this
script asserts.
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
    ap.add_argument("--steps", type=int, default=5, help="timed steps of the traced bench.py run (--steps)")
    ap.add_argument("--warmup", type=int, default=3, help="its warm-up steps (--warmup)")
    ap.add_argument("--marker", default="proposal_decode")
    ap.add_argument("--markers-per-step", type=int, default=1, help="2 for the alternating workload")
    ap.add_argument("--stats-csv", default="", help="write per-kernel totals of the timed steps here")
    args = ap.parse_args()
    rows = load(args.root)
    assert rows, "no kernel trace under %s" % args.root
    marks = [i for i, r in enumerate(rows) if args.marker in r[2]]
    mps = args.markers_per_step
    need = (args.warmup + args.steps) * mps
    assert len(marks) >= need, "marker %r found %d times, expected >= %d" % (args.marker, len(marks), need)
    # step i of the timed region = [marker of timed step i, marker of timed step i + 1): one full step
    # period in steady state; the last timed step has no closing marker and is left out
    first = args.warmup * mps
    bounds = [marks[first + i * mps] for i in range(args.steps)]
    steps = [rows[a:b] for a, b in zip(bounds[:-1], bounds[1:])]
    out = []
    per_kernel = {}
    for st in steps:
        span = st[-1][1] - st[0][0]
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
        for s, e, n in st:
            d = per_kernel.setdefault(n, [0, 0])
            d[0] += 1
            d[1] += e - s
        out.append(dict(span_ms=round(span / 1e6, 3), busy_ms=round(busy / 1e6, 3), idle_ms=round((span - busy) / 1e6, 3),
                        busy_frac=round(busy / span, 4), kernels=len(st),
                        top_gaps=[dict(us=round(g / 1e3, 1), after=a, before=b) for g, a, b in gaps[:5]]))
    naive_timed = sum(c for n, (c, _) in per_kernel.items() if "naive_conv" in n)
    naive_all = sum(1 for _, _, n in rows if "naive_conv" in n)
    res = dict(steps=out, intervals=len(out), naive_conv_kernels_in_timed_steps=naive_timed,
               naive_conv_kernels_in_whole_process=naive_all,
               mean_busy_frac=round(sum(o["busy_frac"] for o in out) / len(out), 4),
               mean_idle_ms=round(sum(o["idle_ms"] for o in out) / len(out), 3),
               mean_span_ms=round(sum(o["span_ms"] for o in out) / len(out), 3))
    if args.stats_csv:
        tot = sum(t for _, t in per_kernel.values())
        with open(args.stats_csv, "w", newline="") as fh:
            w = csv.writer(fh)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "CallsPerStep", "MsPerStep"])
            for n, (c, t) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
                w.writerow([n, c, t, "%.1f" % (t / c), "%.3f" % (100.0 * t / tot), "%.2f" % (c / len(out)),
                            "%.4f" % (t / len(out) / 1e6)])
    print(json.dumps(res, indent=1))
    if naive_timed:
        raise SystemExit("naive_conv_* kernels inside the timed steps: a MIOpen search ran in the timed region")


if __name__ == "__main__":
    main()
