#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE PMC passes of the roofline leg (tools/profile_round.sh) into
profiles/hotpath_traffic.json: mean HBM-side bytes per launch of each RoI-pool kernel, tagged with
the launch shape and the kernel-source id they were taken on (bench.py reports `traffic` only
when both match its own run).

    python3 tools/traffic_json.py <dir with pmc_*> <out.json> <roofline_leg json log>

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE = TCC_EA0_RDREQ x 64 B counts a
128-byte request as 64 B, so wide coalesced reads are under-counted by exactly 2x on gfx950:
read bytes = 2 x FETCH_SIZE x 1024.  WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
The guide calibrated that on 16-byte-per-lane loads; the backward walk loads 8 + 2 bytes per lane,
so the request mix is checked with TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum as well (all of its
requests are 64-byte or larger when the 32B count is ~0) and both figures are recorded."""
import collections, csv, glob, json, os, sys
root, out = sys.argv[1], sys.argv[2]
leg = {}
if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
    for line in open(sys.argv[3]):
        if line.startswith("{"):
            leg = json.loads(line)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wssdl" not in k:
            continue
        name = k.split("(")[0].replace("void ", "").strip()
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / len(v) if v else None
res = {}
for k, cs in acc.items():
    fetch, write = mean(cs.get("FETCH_SIZE", [])), mean(cs.get("WRITE_SIZE", []))
    if fetch is None or write is None:
        continue
    res[k] = dict(launches=len(cs["FETCH_SIZE"]), fetch_size_kb=fetch, write_size_kb=write,
                  read_bytes_corrected=2.0 * fetch * 1024, write_bytes=write * 1024,
                  hbm_bytes_per_launch=2.0 * fetch * 1024 + write * 1024,
                  tcc_hit=mean(cs.get("TCC_HIT_sum", [])), tcc_miss=mean(cs.get("TCC_MISS_sum", [])),
                  ea_rdreq=mean(cs.get("TCC_EA0_RDREQ_sum", [])), ea_rdreq_32b=mean(cs.get("TCC_EA0_RDREQ_32B_sum", [])),
                  ea_wrreq=mean(cs.get("TCC_EA0_WRREQ_sum", [])), ea_wrreq_64b=mean(cs.get("TCC_EA0_WRREQ_64B_sum", [])))
    h, m = res[k]["tcc_hit"], res[k]["tcc_miss"]
    if h is not None and m is not None and h + m > 0:
        res[k]["l2_hit_rate"] = h / (h + m)
meta = leg.get("meta", {})
entry = dict(source="rocprofv3 --pmc passes (one counter group per run) of `python3 tools/roofline_leg.py --iters 5`",
             correction="read = 2 x FETCH_SIZE (gfx950), write = WRITE_SIZE",
             roi_set=leg.get("roi_set"),
             launch={k: meta.get(k) for k in ("N", "H", "W", "C", "R", "argmax_bytes", "kernel_source_id",
                                              "backward_plan", "forward_variant", "backward_variant")},
             kernels=res)
# `--leg NAME` (5th argument): add / replace this launch shape under "legs" of an existing file instead of rewriting it
leg_name = sys.argv[4] if len(sys.argv) > 4 else None
if leg_name and os.path.exists(out):
    doc = json.load(open(out))
    entry["leg"] = leg_name
    legs = [e for e in doc.get("legs", []) if e.get("leg") != leg_name]
    doc["legs"] = legs + [entry]
    json.dump(doc, open(out, "w"), indent=1, sort_keys=True)
else:
    json.dump(entry, open(out, "w"), indent=1, sort_keys=True)
print(out, len(res), "kernels", leg_name or "(top level)")
