import os, sys, subprocess, json
for v in ("0", "1", "2", "3", "8"):
    env = dict(os.environ, WSSDL_ABLATE=v)
    out = subprocess.run([sys.executable, "tools/kernel_bench.py", "--config", "3", "--iters", "10"], env=env, capture_output=True, text=True).stdout
    for l in out.splitlines():
        if "roi_pool_backward" in l:
            d = json.loads(l); print("ablate", v, "ms %.3f" % d["ms"], flush=True)
