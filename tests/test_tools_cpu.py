"""CPU: the measurement helpers that turn profiler output into the committed summaries."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, rows):
    with open(path, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for name, s, e in rows:
            w.writerow(["KERNEL_DISPATCH", name, s * 1000, e * 1000])       # units of the test: microseconds


def test_trace_gaps_timed_steps_only(tmp_path):
    """tools/trace_gaps.py: steps are delimited by the marker kernel, warm-up steps (with MIOpen's naive_conv
    reference kernels) are left out of the per-kernel summary, busy time merges overlapping kernels, and a
    naive_conv kernel INSIDE a timed step fails the script."""
    d = tmp_path / "prof" / "run"
    d.mkdir(parents=True)
    rows, t = [], 0
    for step in range(6):                                   # 2 warm-up + 4 timed
        rows.append(("wssdl::proposal_decode_kernel(float const*)", t, t + 10))
        if step < 2:
            rows.append(("naive_conv_ab_nonpacked_fwd_nhwc_float_double_float", t + 10, t + 5000))
            t += 5000
        rows.append(("backbone_kernel", t + 20, t + 520))    # 10 us gap after the marker
        rows.append(("overlapping_kernel_on_another_queue", t + 100, t + 300))
        rows.append(("wssdl::roi_pool_fwd_rows_kernel<4>", t + 600, t + 1000))   # 80 us gap
        t += 1000
    _trace(str(d / "1_kernel_trace.csv"), rows)
    out_csv = str(tmp_path / "stats.csv")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "trace_gaps.py"), str(tmp_path / "prof"), "--steps", "4",
           "--warmup", "2", "--stats-csv", out_csv]
    res = json.loads(subprocess.check_output(cmd).decode())
    assert res["intervals"] == 3 and res["naive_conv_kernels_in_timed_steps"] == 0
    assert res["naive_conv_kernels_in_whole_process"] == 2
    st = res["steps"][0]
    assert st["kernels"] == 4 and abs(st["busy_ms"] - (10 + 500 + 400) / 1e3) < 1e-9
    assert st["top_gaps"][0]["us"] == 80.0 and "backbone_kernel" in st["top_gaps"][0]["after"]
    assert st["top_gaps"][1]["us"] == 10.0 and abs(st["idle_ms"] - 0.09) < 1e-9
    names = [r["Name"] for r in csv.DictReader(open(out_csv))]
    assert names[0] == "backbone_kernel" and not any("naive" in n for n in names)
    # a naive_conv kernel inside a timed step: non-zero exit
    rows.append(("naive_conv_ab_nonpacked_fwd_nhwc_float_double_float", 12500, 12600))
    _trace(str(d / "1_kernel_trace.csv"), rows)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and b"naive_conv" in p.stderr


def test_bench_byte_models_agree_with_the_roofline_leg():
    """bench.py and tools/roofline_leg.py state the same byte counts: 8 B per pooled element in SURVEY 8(d)'s
    layout, 5 B moved by the 1-byte arg-max pair, plus the feature map once."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import roofline_leg
    m = dict(N=8, H=38, W=63, C=1024, R=8512)
    for op in ("roi_pool_forward", "roi_pool_backward"):
        assert bench.alg_bytes(op, m) == roofline_leg.alg_bytes(op, **m)
        assert roofline_leg.alg_bytes(op, **m) - roofline_leg.moved_bytes(op, **m) == m["R"] * 49 * m["C"] * 3
    assert roofline_leg.moved_bytes("roi_pool_backward", **m) == 2213937152          # the figure in DESIGN.md / profiles


def test_bare_bench_gpus_n_starts_the_launcher_before_torch(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it must start `python -m torch.distributed.run
    --nproc-per-node 2 ... bench.py --gpus 2` as a child (and never initialise torch / HIP itself).  Without a GPU the
    two ranks exit with bench.py's own "needs a GPU" message: the parent relays the non-zero exit code, prints no
    JSON line, and the message proves that rank processes were started with RANK / WORLD_SIZE set."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    err = p.stderr.decode()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:          # on a GPU box this form is covered by tests/test_gpu_distributed.py
        return
    assert p.returncode != 0
    assert not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert "needs a GPU" in err and "torch.distributed" in err.replace("torchrun", "torch.distributed"), err[-1500:]


def test_every_tool_script_parses():
    """tools/*.py and tools/probes/*.py are run on the GPU box only; a syntax error in one of them should show up here."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py")) + glob.glob(os.path.join(root, "tools", "probes", "*.py")))
    assert len(files) >= 20
    for f in files:
        with open(f) as fh:
            ast.parse(fh.read(), filename=f)


def test_owner_model_listing_rule_and_numbers():
    """tools/owner_model.py restates the bin-owner form's listing rule (csrc/roi_pool_walk.hip: axis_entry_own) on the CPU:
    a window is listed once when it fits the region of the tile that holds its first line, otherwise it continues in the
    tile of its first uncovered line; on the fixed roofline set that gives the re-read factors EXPERIMENTS.md quotes."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    import owner_model as om
    assert om.listings(3, 3, 4, 6) == 0                       # empty window
    assert om.listings(3, 6, 4, 6) == 1                       # starts in tile 0 (lines 0-3), region reaches line 5
    assert om.listings(3, 7, 4, 6) == 2                       # line 6 is beyond the region: continues in tile 1
    assert om.listings(0, 16, 4, 6) == 4                      # chain: [0,6) tile 0, [6,10) tile 1, [10,14) tile 2, [14,16) tile 3
    assert om.listings(5, 9, 6, 6) == 2                       # tile == region: every tile the window touches (the exact walk)
    # property: for every window and every (tile, region) the library builds, the pieces partition the window, every piece
    # lies inside the region of the tile that lists it, and no tile lists a window twice
    rs = np.random.RandomState(4)
    for tile, region in ((4, 6), (5, 7), (6, 8), (4, 5), (5, 6), (6, 6), (2, 2), (8, 8)):
        for _ in range(400):
            s0 = int(rs.randint(0, 60))
            e0 = s0 + int(rs.randint(0, 17))
            pieces = om.chain_pieces(s0, e0, tile, region)
            covered = [line for _, a, b in pieces for line in range(a, b)]
            assert covered == list(range(s0, e0)), (s0, e0, tile, region, pieces)
            assert all(t * tile <= a and b <= t * tile + region for t, a, b in pieces)
            assert len({t for t, _, _ in pieces}) == len(pieces)
    rois = np.load(os.path.join(ROOT, "profiles", "roofline_rois_r8512.npy"))
    wins = [om.windows(x) for x in rois]
    exact = om.model(rois, wins, 6, 6, 6, 6)
    own = om.model(rois, wins, 4, 5, 6, 7)
    assert abs(exact["f"] - 1.58) < 0.005 and abs(own["f"] - 1.119) < 0.005
    assert abs(own["traffic_over_moved"] - 1.242) < 0.005     # measured: 1.227 (profiles/hotpath_traffic.json)
