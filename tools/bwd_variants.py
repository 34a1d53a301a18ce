import os, sys, subprocess, json
for v in ("0", "1", "2", "3", "4", "5", "6", "7"):
    env = dict(os.environ, WSSDL_BWD_VARIANT=v)
    out = subprocess.run([sys.executable, "tools/kernel_bench.py", "--config", "3", "--iters", "10"], env=env, capture_output=True, text=True).stdout
    for l in out.splitlines():
        if "roi_pool_backward" in l:
            d = json.loads(l); print("variant", v, "ms %.3f GB/s %.0f" % (d["ms"], d["GBps"]), flush=True)
