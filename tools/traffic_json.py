#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE PMC passes of `bench.py` (tools/profile_round.sh) into
profiles/hotpath_traffic.json: mean HBM-side bytes per launch of each hot-path kernel.

Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE = TCC_EA0_RDREQ x 64 B counts a
128-byte request as 64 B, so wide coalesced reads are under-counted by exactly 2x on gfx950:
read bytes = 2 x FETCH_SIZE x 1024.  WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
(Our backward kernel reads 4 B per lane: the 2x factor is confirmed for it by
TCC_EA0_RDREQ x 128 B matching visits x bytes within 3 %; see DESIGN.md.)"""
import collections, csv, glob, json, os, sys
root = sys.argv[1]
out = sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "wssdl" not in k:
            continue
        name = k.split("(")[0].replace("void ", "").strip()
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, cs in acc.items():
    fetch = sum(cs.get("FETCH_SIZE", [0])) / max(len(cs.get("FETCH_SIZE", [1])), 1)
    write = sum(cs.get("WRITE_SIZE", [0])) / max(len(cs.get("WRITE_SIZE", [1])), 1)
    res[k] = dict(launches=len(cs.get("FETCH_SIZE", [])), fetch_size_kb=fetch, write_size_kb=write,
                  read_bytes_corrected=2.0 * fetch * 1024, write_bytes=write * 1024,
                  hbm_bytes_per_launch=2.0 * fetch * 1024 + write * 1024)
json.dump(dict(source="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py --steps 3 --warmup 2`",
               correction="read = 2 x FETCH_SIZE (gfx950), write = WRITE_SIZE", kernels=res),
          open(out, "w"), indent=1, sort_keys=True)
print(out, len(res), "kernels")
