#!/usr/bin/env python3
"""Prints the numbers the "Reading them" paragraph of profiles/README.md quotes, from the committed
profiles/<tag>_bench_* files (so that the paragraph can be checked, or refreshed, against the files).

    python3 tools/profiles_readme_numbers.py [tag]        (default r04)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"


def line(w):
    return json.loads(open(os.path.join(P, "%s_bench_%s.json.log" % (TAG, w))).read().strip().splitlines()[-1])


def main():
    d = line("resnet50_joint_b8")
    r = d["roofline"]
    fs = r["fixed_set"]
    ks = {}
    for row in csv.DictReader(open(os.path.join(P, TAG + "_bench_resnet50_joint_b8_kernel_stats.csv"))):
        ks[row["Name"].split("(")[0].replace("void ", "").replace("wssdl::", "")] = float(row["MsPerStep"])
    g = json.load(open(os.path.join(P, TAG + "_bench_resnet50_joint_b8_step_gaps.json")))
    print("default: %.1f images/s, %.1f ms/step, hot path %.3f ms (+ %.3f loss ops)" % (
        d["value"], d["ms_per_step"], d["hot_path"]["gpu_ms_per_step"], d["hot_path"]["loss_op_ms_per_step"]))
    print("roofline: backward %.4f ms frac %.3f frac_8d %.3f frac_traffic %.3f; forward %.4f ms" % (
        fs["roi_pool_backward"]["avg_ms"], r["frac"], r["frac_8d"], r["frac_traffic"] or float("nan"),
        fs["roi_pool_forward"]["avg_ms"]))
    for prefix in ("roi_pool_fwd_rows", "roi_pool_bwd_walk", "nms_mask_sweep_fused", "proposal_decode_runs", "order_rank",
                   "roi_sample", "mtl_backward", "anchor_subsample"):
        print("  in-step %-24s %.4f ms" % (prefix, sum(v for n, v in ks.items() if n.startswith(prefix))))
    for k in ("proposal_layer", "proposal_target_layer", "anchor_target_layer", "roi_pool_backward_prepare"):
        print("  between events %-26s %.4f ms" % (k, r["per_kernel"][k]["avg_ms"]))
    print("busy %.1f %%, idle %.2f ms of %.0f" % (100 * g["mean_busy_frac"], g["mean_idle_ms"], g["mean_span_ms"]))
    for w in ("resnet18_sup_b2", "resnet50_alter", "resnet101_1600_test", "vgg16_joint"):
        o = line(w)
        pk = o["roofline"].get("per_kernel", {})
        print("%s: %.1f images/s, %.1f ms/step, hot path %.3f ms, proposal layer %.3f ms" % (
            w, o["value"], o["ms_per_step"], o["hot_path"]["gpu_ms_per_step"], pk.get("proposal_layer", {}).get("avg_ms", 0.0)))


if __name__ == "__main__":
    main()
