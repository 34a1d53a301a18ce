#!/usr/bin/env python3
"""Headline benchmark: images/sec of a full train step (600x1000 synthetic inputs) with the
detection hot path on the HIP library, plus the hot path's roofline and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one optimiser step of the workload on this rank's images (weak scaling: every
rank runs the same per-GPU batch; gradients are averaged with one RCCL all-reduce).  W
untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; the time is
the max over ranks; rank 0 prints ONE JSON line.

`roofline` is measured on a FIXED RoI set (tools/roofline_leg.py: exactly n_sup*128 + n_ws*2000 =
8512 rows for the default workload, profiles/roofline_rois_r8512.npy) right after the timed steps,
with HIP events on the launch stream around each C-ABI call; `cpu_baseline` pools the same RoI set.

Workloads (BASELINE.json `configs`):
  resnet50_joint_b8   configs[2] -- ResNet-50, combined mini-batch (train.py), 4 supervised +
                      4 weak images per GPU: R = 4*128 + 4*2000 RoIs through RoI pool (default:
                      the largest single-GPU 600x1000 train configuration)
  resnet18_sup_b2     configs[1] -- ResNet-18, batch 2 fully supervised
  resnet50_alter      configs[3] -- ResNet-50, one alternating iteration (train_alter.py):
                      a supervised step on 1 image then a weak step on 2 images
  resnet101_1600_test configs[4] -- ResNet-101, 1000x1600, test-mode RPN (6000 -> 300), forward only
  vgg16_joint         VGG-16 combined mini-batch 1 + 2 (the reference's default sizes)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# plumbing: the per-RoI head is fp32 GEMMs; hipBLASLt picks better kernels for these shapes than
# the default BLAS path (measured 192 -> 176 ms per step).  Must be set before torch is imported.
os.environ.setdefault("TORCH_BLAS_PREFER_HIPBLASLT", "1")


def seed_miopen_db():
    """plumbing: the backbone's convolutions run on MIOpen.  Without a search MIOpen's immediate mode picks
    its solvers by heuristic (163 ms per step); with `torch.backends.cudnn.benchmark` it times the candidates
    of every convolution once (a 'find': 138-148 ms per step, but ~2.5 minutes of search and kernel builds
    on a fresh machine).  wssdl_bus_amd/miopen_db/ holds the RESULT of that search for the bench workloads
    on MI355X -- MIOpen's own user find-db (text: problem -> solvers and their times) and the code objects of
    the chosen kernels -- so that the search is a look-up.  They are copied to a scratch directory (MIOpen
    writes to its user db) unless the caller already points MIOpen somewhere.
    FAILS CLOSED: the db is named after the MIOpen build that wrote it (gfx950100.HIP.<version>-<hash>); when
    the MIOpen library this process will load does not carry that very tag, MIOpen would ignore the db and
    search for ~2.5 minutes on every rank -- so the function returns (None, reason) and main() turns the
    find mode off (heuristic solver choice).  The scratch copy is a fresh private directory (mkdtemp, 0700),
    removed at exit.  Returns (scratch dir or None, reason)."""
    if os.environ.get("MIOPEN_USER_DB_PATH") or os.environ.get("MIOPEN_CUSTOM_CACHE_DIR"):
        return None, "caller set MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR"
    src = os.path.join(ROOT, "wssdl_bus_amd", "miopen_db")
    tags = [f[len("gfx950100."):-len(".ufdb.txt")] for f in (os.listdir(os.path.join(src, "config"))
            if os.path.isdir(os.path.join(src, "config")) else []) if f.startswith("gfx950100.") and f.endswith(".ufdb.txt")]
    if not tags:
        return None, "no find-db shipped"
    import importlib.util
    import mmap
    spec = importlib.util.find_spec("torch")
    lib = os.path.join(os.path.dirname(spec.origin), "lib", "libMIOpen.so") if spec and spec.origin else ""
    try:
        with open(lib, "rb") as fh:
            mm = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
            hit = mm.find(tags[0].encode()) >= 0
            mm.close()
    except (OSError, ValueError):
        hit = False
    if not hit:
        return None, "shipped find-db is for MIOpen build %s, which %s does not identify as" % (tags[0], lib or "torch's MIOpen")
    import atexit
    import shutil
    import tempfile
    dst = tempfile.mkdtemp(prefix="wssdl_miopen_db_")           # private (0700), unpredictable name
    atexit.register(shutil.rmtree, dst, True)
    try:
        for sub in ("config", "cache"):
            if os.path.isdir(os.path.join(src, sub)):
                shutil.copytree(os.path.join(src, sub), os.path.join(dst, sub))
    except OSError as e:
        return None, "could not copy the find-db: %s" % e
    os.environ["MIOPEN_USER_DB_PATH"] = os.path.join(dst, "config")
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = os.path.join(dst, "cache")
    return dst, "wssdl_bus_amd/miopen_db (%s)" % tags[0]


def miopen_db_grew(scratch):
    """True when MIOpen added find results to the scratch copy, i.e. it had to search (db miss)."""
    if not scratch:
        return None
    src = os.path.join(ROOT, "wssdl_bus_amd", "miopen_db", "config")
    try:
        for f in os.listdir(os.path.join(scratch, "config")):
            a = os.path.join(scratch, "config", f)
            b = os.path.join(src, f)
            if not os.path.exists(b) or os.path.getsize(a) != os.path.getsize(b):
                return True
    except OSError:
        return None
    return False

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured achievable

WORKLOADS = {
    "resnet50_joint_b8": dict(net="Resnet_train", depth=50, mode="joint", n_sup=4, n_ws=4,
                              im=(600, 1000), baseline_config=2, fixed_rois=True),
    # (round 6: every workload's roofline leg runs on a COMMITTED set of its own network's proposals when the file
    # exists -- saved once with --save-fixed-rois -- so that the PMC passes in profiles/hotpath_traffic.json belong to
    # exactly the launches the line reports; without the file the set is built from this run's network)
    "resnet18_sup_b2": dict(net="Resnet_train_alter", depth=18, mode="sup", n_sup=2, n_ws=0,
                            im=(600, 1000), baseline_config=1, fixed_rois="roofline_rois_resnet18_sup_b2_r256.npy"),
    "resnet50_alter": dict(net="Resnet_train_alter", depth=50, mode="alter", n_sup=1, n_ws=2,
                           im=(600, 1000), baseline_config=3, fixed_rois="roofline_rois_resnet50_alter_weak_r4000_large.npy"),
    "resnet101_1600_test": dict(net="Resnet_train", depth=101, mode="test", n_sup=1, n_ws=0,
                                im=(1000, 1600), baseline_config=4, fixed_rois="roofline_rois_resnet101_1600_test_r300.npy"),
    "vgg16_joint": dict(net="VGGnet_train", depth=16, mode="joint", n_sup=1, n_ws=2,
                        im=(600, 1000), baseline_config=0, fixed_rois="roofline_rois_vgg16_joint_r4128.npy"),
}


def alg_bytes(name, m):
    """Algorithmic bytes of one launch (restated in DESIGN.md)."""
    if name == "roi_pool_forward":
        return m["N"] * m["H"] * m["W"] * m["C"] * 4 + m["R"] * 20 + m["R"] * 49 * m["C"] * 8
    if name == "roi_pool_backward":
        return m["R"] * 49 * m["C"] * 8 + m["N"] * m["H"] * m["W"] * m["C"] * 4
    if name == "anchor_target_layer":
        return m["n_out"] * 13 * 9 * m["H"] * m["W"] * 4
    if name == "proposal_layer":
        ka = m["H"] * m["W"] * m["A"]
        nms = m["pre"] * 20 + m["pre"] * ((m["pre"] + 63) // 64) * 8
        return m["N"] * (ka * 6 * 4 + ka * 5 * 4 + 2 * ka * 8 + nms)
    return 0


def cpu_baseline(wl, fixed_rois=None, seed=3):
    """The oracle (a port of the reference's CPU path; its NMS/IoU inner kernels are the
    reference's own Cython when oracle/_ref is present) timed on this host:
      layers   anchor targets + proposal layer + proposal targets of every image of the per-GPU
               mini-batch (each with its own synthetic RPN maps), single-threaded like the reference
               (py_func under the GIL);
      RoI pool forward + backward on the SAME fixed RoI set the GPU roofline leg uses, on all host
               cores (forward sharded over RoIs like the reference's Shard(), roi_pooling_op.cc:198-203;
               backward sharded over channels: the reference shards the flat bottom range,
               :460-465).
    One pass is 10-30 s of CPU work; repeated (at most 3 passes) until 10 s have been measured."""
    import numpy as np
    from oracle import c_oracle, np_oracle as O, ref_kernels
    H = -(-wl["im"][0] // 16)
    W = -(-wl["im"][1] // 16)
    C = {18: 256, 16: 512}.get(wl["depth"], 1024)
    train = wl["mode"] != "test"
    info = np.array([[wl["im"][0], wl["im"][1], 1.0, 1]], np.float32)
    gt = np.zeros((1, 20, 5), np.float32)
    gt[0, 0] = [100, 80, 380, 300, 1]
    gt[0, 1] = [500, 60, 900, 420, 0]
    ng = np.array([2], np.int32)

    def inputs(k):
        rs = np.random.RandomState(seed + k)
        logits = rs.normal(size=(1, H, W, 9, 2)).astype(np.float32)
        e = np.exp(logits - logits.max(-1, keepdims=True))
        p = (e / e.sum(-1, keepdims=True)).astype(np.float32)
        prob = np.concatenate((p[..., 0], p[..., 1]), axis=-1)
        pred = rs.normal(0, 0.2, size=(1, H, W, 36)).astype(np.float32)
        return prob, pred

    use_ref = ref_kernels.available()
    if use_ref:                       # the reference's own Cython kernels (PyObject-compare NMS)
        O_nms, O_iou, O_ui = O.nms, O.bbox_overlaps, O.bbox_overlaps_ui
        O.nms = lambda d, t: [] if d.shape[0] == 0 else ref_kernels.cpu_nms(d, t)
        O.bbox_overlaps = lambda b, q: ref_kernels.bbox_overlaps(
            np.ascontiguousarray(b[:, :4], dtype=np.float64), np.ascontiguousarray(q[:, :4], dtype=np.float64))
        O.bbox_overlaps_ui = lambda b, q: ref_kernels.bbox_overlaps_ui(
            np.ascontiguousarray(b[:, :4], dtype=np.float64), np.ascontiguousarray(q[:, :4], dtype=np.float64))
    cores = os.cpu_count() or 1
    n_img = wl["n_sup"] + wl["n_ws"]
    t_layers = t_pool = 0.0
    passes = 0
    layer_rois = 0
    rs = np.random.RandomState(seed)
    feat = np.maximum(rs.normal(size=(n_img, H, W, C)), 0).astype(np.float32)
    try:
        while passes < 3 and (passes == 0 or t_layers + t_pool < 10.0):
            layer_rois = 0
            generated = []
            for k in range(n_img):
                weak = k >= wl["n_sup"]
                prob, pred = inputs(passes * n_img + k)
                t0 = time.perf_counter()
                if train and not weak:
                    O.anchor_target_layer(np.zeros((1, H, W, 18), np.float32), gt, ng, info, None, [16], [8, 16, 32],
                                          "SNUBH", rng=np.random.RandomState(seed + k))
                rois = O.proposal_layer(prob, pred, info, train, False, [16], [8, 16, 32])
                if train and not weak:
                    rois = O.proposal_target_layer(rois, gt, ng, 3, True, False,
                                                   rng=np.random.RandomState(seed + k))[0]
                t_layers += time.perf_counter() - t0
                layer_rois += rois.shape[0]
                rois = rois.copy()
                rois[:, 0] = k
                generated.append(rois)
            pool_rois = fixed_rois if fixed_rois is not None else np.concatenate(generated).astype(np.float32)
            t0 = time.perf_counter()
            top, arg = c_oracle.roi_pool_forward(feat, pool_rois, 7, 7, 1.0 / 16, "cuda", threads=cores)
            if train:
                c_oracle.roi_pool_backward(top, arg, pool_rois, feat.shape, 7, 7, 1.0 / 16, threads=cores)
            t_pool += time.perf_counter() - t0
            passes += 1
    finally:
        if use_ref:
            O.nms, O.bbox_overlaps, O.bbox_overlaps_ui = O_nms, O_iou, O_ui
    t_step = (t_layers + t_pool) / passes
    return dict(
        value=n_img / t_step, unit="images/s (hot path only: anchor targets + proposal layer + "
                                   "proposal targets + RoI pool fwd/bwd; no backbone)",
        cores=cores, kind="port",
        sample="%d pass(es) over the per-GPU mini-batch of this workload (%d supervised + %d weak images), "
               "%.1f s of CPU work: layers %.1f s single-threaded like the reference (GIL; they produced %d RoIs "
               "per pass), RoI pool forward + backward %.1f s on %d host threads over R = %d RoIs%s, C = %d; "
               "NMS/IoU inner kernels = %s"
               % (passes, wl["n_sup"], wl["n_ws"], t_layers + t_pool, t_layers, layer_rois, t_pool, cores,
                  pool_rois.shape[0], " (the fixed set of the GPU roofline leg)" if fixed_rois is not None else "",
                  C, "the reference's own Cython (oracle/_ref)" if use_ref else "C port (oracle/_ref absent)"),
        roi_pool_rois=int(pool_rois.shape[0]),
        cpu_roi_pool_ms=t_pool / passes * 1e3, cpu_layers_ms=t_layers / passes * 1e3,
        cpu_hot_path_ms_per_step=t_step * 1e3)


def fixed_roi_set(wl, net, blobs, weak_step=False):
    """The RoI set of the roofline leg: the committed file for the default workload, otherwise
    n_sup * 128 sampled + n_ws * post_nms_topN proposals built from the network like
    tools/make_roofline_rois.py does.  `weak_step`: the alternating mode's weak step (all images weak)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import roofline_leg
    if wl.get("fixed_rois") is True and os.path.exists(roofline_leg.ROIS_PATH):
        return roofline_leg.load_rois()
    if isinstance(wl.get("fixed_rois"), str) and os.path.exists(os.path.join(ROOT, "profiles", wl["fixed_rois"])) \
            and not os.environ.get("WSSDL_BENCH_GENERATE_ROIS"):
        return roofline_leg.load_rois(os.path.join(ROOT, "profiles", wl["fixed_rois"]))
    import torch
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded
    train = wl["mode"] != "test"
    with torch.no_grad():
        was = net.training
        net.train(train)
        L = net(blobs["data"], blobs["im_info"], blobs["gt_boxes"], blobs["num_gt_boxes"], is_training=train,
                is_ws=bool(weak_step) or (wl["mode"] == "alter" and wl["n_sup"] == 0), test_net=not train)
        net.train(was)
    post = int((cfg.TRAIN if train else cfg.TEST).RPN_POST_NMS_TOP_N)
    rp, cnt, dec, sidx, scnt = [t.cpu().numpy() for t in proposal_layer_padded(
        L["rpn_cls_score"], L["rpn_bbox_pred"], blobs["im_info"], train, debug=True, from_logits=True)]
    rows = []
    n_valid = L["roi-data"][1].shape[0] if (train and isinstance(L["roi-data"], tuple) and not weak_step) else 0
    sampled = L["roi-data"][0][:n_valid].cpu().numpy() if n_valid else np.zeros((0, 5), np.float32)
    n_img = blobs["data"].shape[0]
    for i in range(n_img):
        r = sampled[sampled[:, 0] == i]
        if r.shape[0]:
            rows.append(r)
            continue
        kept = rp[i, :cnt[i], 1:]
        cand = dec[i][sidx[i, :scnt[i]]]
        seen = set(map(bytes, np.ascontiguousarray(kept)))
        extra = np.array([c for c in cand if bytes(np.ascontiguousarray(c)) not in seen][:post - len(kept)],
                         dtype=np.float32).reshape(-1, 4)
        boxes = np.concatenate([kept, extra])
        rows.append(np.concatenate([np.full((len(boxes), 1), i, np.float32), boxes], axis=1))
    rois = np.ascontiguousarray(np.concatenate(rows).astype(np.float32))
    return rois, "generated from this run's network (%d rows)" % rois.shape[0]


def hbm_traffic(kernel_key, meta):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/hotpath_traffic.json, tools/profile_round.sh) -- only when they were taken on the same
    kernel sources and the same launch shape as this run; otherwise null."""
    tpath = os.path.join(ROOT, "profiles", "hotpath_traffic.json")
    if not os.path.exists(tpath):
        return None, "no PMC passes committed"
    tj = json.load(open(tpath))
    want = {k: meta[k] for k in ("N", "H", "W", "C", "R", "argmax_bytes", "kernel_source_id")}
    # one entry per launch shape (round 6: the default workload's at the top level as before, the other workloads' legs
    # under "legs"); an entry counts when shape, RoI count and kernel sources are this run's
    entries = [tj] + list(tj.get("legs", []))
    entry = next((e for e in entries if all(e.get("launch", {}).get(k) == v for k, v in want.items())), None)
    if entry is None:
        shapes = [{k: e.get("launch", {}).get(k) for k in ("N", "C", "R", "kernel_source_id")} for e in entries]
        return None, "stale: profiles/hotpath_traffic.json holds %s, this run is %s" % (shapes, want)
    # an op of several kernels: their bytes add up (bin-owner backward = walk + halo merge; split backward = walk +
    # combine; block-table forward = tables + bin-row order + pooling)
    keys = {"roi_pool_bwd": ("roi_pool_bwd_walk_kernel", "walk_merge_kernel", "walk_merge_split_kernel", "walk_combine_kernel"),
            "roi_pool_fwd": ("roi_pool_fwd_rows_kernel", "roi_pool_fwd_compact_kernel", "roi_pool_fwd_blocks_kernel",
                             "blocks_build_kernel", "rows_scatter_kernel")}.get(kernel_key, (kernel_key,))
    hit = [(k, v) for k, v in entry.get("kernels", {}).items() if any(q in k for q in keys)]
    if hit:
        return int(sum(v["hbm_bytes_per_launch"] for _, v in hit)), "profiles/hotpath_traffic.json (%s)" % " + ".join(k for k, _ in hit)
    return None, "kernel not in profiles/hotpath_traffic.json"


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the launcher the scaling run
    uses (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`) as a CHILD process,
    hand on rank 0's one JSON line and the exit code.  Called before torch / HIP are imported: this process
    never touches the GPU, and nothing is exec'ed over a process that did."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, universal_newlines=True)
    lines = []
    for line in proc.stdout:
        is_result = False
        if line.lstrip().startswith("{"):
            try:
                is_result = "metric" in json.loads(line)
            except ValueError:
                pass
        if is_result:
            lines.append(line.rstrip("\n"))
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    for line in lines[-1:]:
        print(line, flush=True)
    return rc if rc else (0 if lines else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="resnet50_joint_b8", choices=sorted(WORKLOADS))
    ap.add_argument("--sampling-rng", default="device", choices=["device", "reference"],
                    help="anchor sub-sampling RNG: 'device' (no host round trip) or the reference's numpy stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-miopen-benchmark", action="store_true",
                    help="do not let MIOpen time its candidate solvers for the (static-shape) trunk convolutions "
                         "(torch.backends.cudnn.benchmark off: heuristic solver choice)")
    ap.add_argument("--miopen-benchmark", action="store_true",
                    help="find mode even without a matching shipped find-db (searches: minutes on a fresh machine)")
    ap.add_argument("--no-fused-rpn-softmax", action="store_true",
                    help="materialise rpn_cls_prob with separate reshape / softmax / reshape ops instead of "
                         "fusing them into the proposal decode kernel (f2, the default)")
    ap.add_argument("--no-fused-loss", action="store_true",
                    help="the four supervised loss terms as the chain of torch ops instead of the one device op "
                         "(a13, csrc/loss.hip, the default)")
    ap.add_argument("--roofline-iters", type=int, default=20, help="launches of the fixed-RoI roofline leg")
    ap.add_argument("--padded-rois", action="store_true",
                    help="fixed-shape RoI blob (dead rows carry batch index -1): no device->host copy between "
                         "the backbone and the loss; the per-RoI head then runs on the padded row count")
    ap.add_argument("--save-fixed-rois", default="", metavar="PATH.npy",
                    help="write the RoI set of the roofline leg (float32 [R,5]) to this file (lab use: the sets of "
                         "the workloads other than the default are built from the run's own network)")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=INT",
                    help="wssdl_set_tuning(KEY, INT) before the run, for A/B comparisons (e.g. nms_fused=0); repeatable")
    ap.add_argument("--roi-bwd-plan", type=int, default=-1,
                    help="force a plan of the list-driven RoI-pool backward (wssdl_set_tuning roi_bwd_plan; -1 = automatic)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # before MIOpen initialises.  No usable find-db -> no find mode (a search costs minutes per rank)
    miopen_db, miopen_db_note = (None, "--no-miopen-benchmark") if args.no_miopen_benchmark else seed_miopen_db()
    if miopen_db is None and not args.miopen_benchmark and not os.environ.get("MIOPEN_USER_DB_PATH"):
        args.no_miopen_benchmark = True
    import numpy as np
    import torch
    from wssdl_bus_amd import _lib, synthetic
    from wssdl_bus_amd.distributed import DistContext
    from wssdl_bus_amd.fast_rcnn.config import cfg
    from wssdl_bus_amd.fast_rcnn.train_bus import SolverWrapper
    from wssdl_bus_amd.networks.factory_bus import get_network

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the HIP hot path has no CPU fallback)")
    torch.backends.cudnn.benchmark = not args.no_miopen_benchmark
    _lib.lib()
    if args.roi_bwd_plan >= 0:
        _lib.set_tuning("roi_bwd_plan", args.roi_bwd_plan)
    for kv in args.tuning:
        key, _, val = kv.partition("=")
        _lib.set_tuning(key, int(val))
    ctx = DistContext()
    if ctx.world_size != args.gpus:
        if ctx.world_size == 1 and args.gpus > 1:
            sys.exit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(ctx.local_rank % torch.cuda.device_count())
    wl = WORKLOADS[args.workload]
    cfg.TRAIN.IMS_PER_BATCH = wl["n_sup"]
    cfg.TRAIN.WS_IMS_PER_BATCH = wl["n_ws"]
    cfg.SAMPLING_RNG = args.sampling_rng
    cfg.FUSED_RPN_SOFTMAX = not args.no_fused_rpn_softmax
    cfg.FUSED_LOSS = not args.no_fused_loss
    cfg.PADDED_ROIS = bool(args.padded_rois)
    seed = ctx.seed(cfg.RNG_SEED)
    cfg.DEVICE_RNG_SEED = int(cfg.RNG_SEED)  # base; SolverWrapper adds the rank when data-parallel
    np.random.seed(seed)
    torch.manual_seed(seed)

    net = get_network(wl["net"], wl["depth"]).cuda().to(memory_format=torch.channels_last)
    # identical initial weights on every rank (rank 0's), as a data-parallel job needs
    if ctx.enabled:
        for p in net.state_dict().values():
            torch.distributed.broadcast(p, 0)
    solver = SolverWrapper(net, dist_ctx=ctx)
    im_h, im_w = wl["im"]
    mode = wl["mode"]
    if mode == "alter":
        blobs_s = synthetic.make_batch(wl["n_sup"], 0, im_h, im_w, seed)
        blobs_ws = synthetic.make_batch(0, wl["n_ws"], im_h, im_w, seed + 1000)
    else:
        blobs = synthetic.make_batch(wl["n_sup"], wl["n_ws"], im_h, im_w, seed)
    images_per_step = wl["n_sup"] + wl["n_ws"]

    def step():
        if mode == "joint":
            return solver.train_step_joint(blobs)
        if mode == "alter":
            return solver.train_step_alter(blobs_s, blobs_ws)
        if mode == "sup":
            layers = net(blobs["data"], blobs["im_info"], blobs["gt_boxes"], blobs["num_gt_boxes"],
                         is_training=True, is_ws=False)
            from wssdl_bus_amd.fast_rcnn.train_bus import supervised_loss
            losses = supervised_loss(layers, net.weight_decay_params())
            losses["loss"].backward()
            solver._apply()
            return losses
        # test: im_detect (test_bus.py:146-240) + the post-detection step of test_net (:360-401): per-class
        # score threshold, NMS at cfg.TEST.NMS on the HIP kernel, max_per_image cap (f3)
        from wssdl_bus_amd.fast_rcnn.test_bus import im_detect, postprocess_detections
        scores, boxes = im_detect(net, blobs["data"], blobs["im_info"])
        with _lib.timed("postprocess_detections", dict(R=int(scores.shape[0]), classes=int(scores.shape[1]))):
            return postprocess_detections(scores, boxes, scores.shape[1], thresh=0.05, max_per_image=300)

    net.train()
    for _ in range(args.warmup):
        step()
    _lib.timeline.reset(True)
    if getattr(solver, "overlap", None) is not None:
        solver.overlap.start_timing()      # exposed all-reduce wait per optimiser step (scaling_diag)
    ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    _lib.timeline.enabled = False
    # what a single SCALE line needs to explain a miss: every rank's own step time, the all-reduce wait backward did
    # not hide (per optimiser step, per rank), the buckets and their bytes, buckets launched only after backward
    my_ms = elapsed / max(args.steps, 1) * 1e3
    ov = getattr(solver, "overlap", None)
    ov_stats = ov.stats() if ov is not None else None
    scaling_diag = dict(
        ms_per_step_per_rank=[round(v, 3) for v in ctx.gather_floats(my_ms)],
        exposed_allreduce_ms_per_optimizer_step_per_rank=[round(v, 4) for v in ctx.gather_floats(
            ov_stats["exposed_wait_ms_per_finish"] if ov_stats else 0.0)],
        exposed_allreduce_ms_max_per_rank=[round(v, 4) for v in ctx.gather_floats(
            ov_stats["exposed_wait_ms_max"] if ov_stats else 0.0)],
        optimizer_steps_per_step=(round(ov_stats["finishes"] / max(args.steps, 1), 2) if ov_stats else None),
        buckets=(ov_stats["buckets"] if ov_stats else 0), bucket_bytes=(ov_stats["bucket_bytes"] if ov_stats else []),
        bucket_bytes_cap=(ov_stats["bucket_bytes_cap"] if ov_stats else None),
        allreduce_bytes_per_optimizer_step=(ov_stats["allreduce_bytes_per_finish"] if ov_stats else 0),
        late_buckets_per_optimizer_step=(round(ov_stats["late_buckets_per_finish"], 2) if ov_stats else None),
        backend=(ctx.backend if ctx.enabled else None),
        note="exposed = the compute stream's wait in GradOverlap.finish() for collectives backward did not hide "
             "(events around the waits); late buckets = launched by finish(), never overlapped; one process per GPU, "
             "image-parallel, no data-path collective")
    elapsed = ctx.max_over_ranks(elapsed)
    # the deferred device flags of the timed steps (RoI-pool window / list overflow, NMS time-out): the polls
    # inside the steps run one step late and never see the last one
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op
    roi_pooling_op.check_flags()
    n_ranks_seen = ctx.count_ranks()      # a sum of ones over the job's collective backend
    loss_val = float(out["loss"].detach()) if isinstance(out, dict) and torch.is_tensor(out.get("loss")) else None

    def measured_copy_gbps():
        # device-to-device copy of 1 GiB (read + write counted), the practical HBM ceiling
        # to read `roofline.frac` against
        a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
        b = torch.empty_like(a)
        b.copy_(a)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5):
            b.copy_(a)
        e.record()
        torch.cuda.synchronize()
        return 5 * 2 * a.numel() * 4 / (s.elapsed_time(e) * 1e-3) / 1e9

    # ---- roofline leg: the RoI-pool pair of the training path on the FIXED RoI set (every rank
    # runs it so that ranks stay in step; rank 0 reports)
    tl = _lib.timeline.summary()          # hot-path launches of the timed region (this rank)
    feat_key = "conv5_3" if wl["net"].startswith("VGG") else "group2/relu"
    # (alternating mode: the leg measures the WEAK step's launch -- all of its images' proposals through RoI pooling,
    # the heavier of the iteration's two -- not the supervised step's 128 rows)
    leg_blobs = blobs_ws if mode == "alter" else blobs
    rois_fixed, roi_tag = fixed_roi_set(wl, net, leg_blobs, weak_step=(mode == "alter"))
    if args.save_fixed_rois and ctx.rank == 0:
        np.save(args.save_fixed_rois, rois_fixed)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import roofline_leg
    N_leg = int(leg_blobs["data"].shape[0])
    H_leg, W_leg, C_leg = [int(v) for v in net.layers[feat_key].shape[1:4]]
    if rois_fixed.shape[0] == 0 or int(rois_fixed[:, 0].max()) >= N_leg:
        N_leg = int(rois_fixed[:, 0].max()) + 1 if rois_fixed.shape[0] else N_leg
    leg, leg_meta = roofline_leg.run(rois_fixed, N_leg, H_leg, W_leg, C_leg, iters=args.roofline_iters)

    if ctx.rank == 0:
        # hot_ms: the ops the CPU baseline also runs (layers + RoI pool); the fused loss op (a13) is
        # listed beside it -- its CPU counterpart is the TF graph, which the baseline does not time
        loss_ops = ("multi_task_loss", "multi_task_loss_backward", "mil_loss", "mil_loss_backward")
        hot_ms = sum(d["total_ms"] for k, d in tl.items() if k not in loss_ops) / max(args.steps, 1)
        loss_ms = sum(d["total_ms"] for k, d in tl.items() if k in loss_ops) / max(args.steps, 1)
        # dominant kernel = the single-kernel launch of the leg with the largest average duration
        # (roi_pool_backward = the walk kernel alone; its list-building prepare step, like the
        # proposal / target layers, is a chain of small latency-bound kernels: per_kernel only)
        # (bin-owner form of the backward: "roi_pool_backward" = its walk + the halo merge, two kernels, one op)
        single = [k for k in ("roi_pool_forward", "roi_pool_backward") if k in leg]
        dom = max(single, key=lambda k: leg[k]["avg_ms"])
        d = leg[dom]
        kernel_key = {"roi_pool_forward": "roi_pool_fwd", "roi_pool_backward": "roi_pool_bwd"}[dom]
        traffic, traffic_src = hbm_traffic(kernel_key, leg_meta)
        # Three byte counts per launch of the dominant kernel, one duration:
        #   moved   what THIS implementation has to move at least (1-byte arg-max: 5 B per pooled element):
        #           the figure `achieved` / `frac` are quoted on;
        #   8d      SURVEY.md 8(d)'s figure for the reference's layout (f32 top + i32 arg-max, 8 B per
        #           element): what a launch in the reference's layout would have moved -- `frac_8d`;
        #   traffic HBM-side bytes from the committed PMC passes of the same launch -- `frac_traffic`.
        secs = d["avg_ms"] * 1e-3
        moved = int(d.get("min_moved_bytes", d["alg_bytes_per_launch"]))
        helpers = {"roi_pool_forward": ("roi_pool_forward_windows",),
                   "roi_pool_backward": ("roi_pool_backward_prepare",)}[dom]
        helper_ms = sum(leg[h]["avg_ms"] for h in helpers if h in leg)
        roofline = dict(
            bound="hbm", kernel=dom, achieved=round(moved / secs / 1e9, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
            frac=round(moved / secs / 1e9 / HBM_PEAK_GBPS, 4), traffic=traffic, traffic_source=traffic_src,
            frac_moved=round(moved / secs / 1e9 / HBM_PEAK_GBPS, 4),
            frac_traffic=(round(traffic / secs / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None),
            frac_8d=round(d["GBps"] / HBM_PEAK_GBPS, 4), achieved_8d=round(d["GBps"], 1),
            traffic_over_moved=(round(traffic / moved, 3) if traffic else None),
            moved_bytes_per_launch=moved, alg_bytes_8d_per_launch=int(d["alg_bytes_per_launch"]),
            R=int(rois_fixed.shape[0]), roi_set=roi_tag, launch=leg_meta,
            avg_launch_ms=round(d["avg_ms"], 4), helper_launches_ms=round(helper_ms, 4),
            frac_moved_with_helpers=round(moved / ((d["avg_ms"] + helper_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            launches=d["calls"], measured_d2d_copy=round(measured_copy_gbps(), 1),
            note="achieved / frac = bytes this implementation must move (f32 top_diff + 1-byte arg-max codes + "
                 "bottom_diff once) / launch time of the dominant op alone (the slower of the RoI-pool pair; the backward's "
                 "default form since round 5 is the bin-owner walk = walk kernel + halo merge, see launch.backward_variant, "
                 "with the exact walk kept as the bit-for-bit parity reference); its helper launches (window "
                 "table / list building, latency-bound chains) are in helper_launches_ms and "
                 "frac_moved_with_helpers.  frac_8d uses SURVEY.md 8(d)'s bytes for the reference's f32 + i32 "
                 "layout.  RoI-pool parity is pinned by hand-computed cases and two independent oracle "
                 "restatements, not by reference outputs (TensorFlow op cannot be built here: 'parity unpinned', "
                 "DESIGN.md section 2)",
            # both kernels of the pair (round 5: within a few per cent of each other, so which one is "dominant" can
            # change from box to box): launch time, fraction of peak on moved bytes, HBM-side bytes of the PMC passes
            pair={k: dict(avg_ms=round(leg[k]["avg_ms"], 4),
                          frac_moved=round(leg[k].get("min_moved_bytes", leg[k]["alg_bytes_per_launch"]) / (leg[k]["avg_ms"] * 1e-3)
                                           / 1e9 / HBM_PEAK_GBPS, 4),
                          traffic=hbm_traffic({"roi_pool_forward": "roi_pool_fwd", "roi_pool_backward": "roi_pool_bwd"}[k], leg_meta)[0],
                          kernels=(2 if (k == "roi_pool_backward" and (leg_meta.get("backward_owner_plan") or -1) >= 0) else 1))
                  for k in single},
            fixed_set={k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                       for k, v in leg.items()},
            per_kernel={k: dict(avg_ms=round(v["avg_ms"], 4), calls=v["calls"],
                                kernels=(1 if k in ("roi_pool_forward", "roi_pool_backward") else "chain"),
                                GBps=round(sum(alg_bytes(k, m) for m in v["metas"]) / v["calls"]
                                           / (v["avg_ms"] * 1e-3) / 1e9, 1))
                        for k, v in tl.items()})
        result = {
            "metric": "images/sec (train step, 600x1000)" if mode != "test" else "images/sec (test forward, 1000x1600)",
            "value": round(images_per_step * ctx.world_size * args.steps / elapsed, 3),
            "unit": "images/s",
            "n_gpus": ctx.world_size, "n_ranks_seen": n_ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "baseline_config_index": wl["baseline_config"],
                       "backbone": "%s-%d" % ("ResNet" if wl["net"].startswith("Resnet") else "VGG", wl["depth"]),
                       "image": "%dx%d" % (im_h, im_w), "images_per_gpu": images_per_step,
                       "supervised_per_gpu": wl["n_sup"], "weak_per_gpu": wl["n_ws"], "mode": mode,
                       "parallelism": "image-parallel dp%d, %s grad all-reduce" % (
                           ctx.world_size, "RCCL" if (ctx.backend or "nccl") == "nccl" else ctx.backend),
                       "sampling_rng": args.sampling_rng, "fused_rpn_softmax": bool(cfg.FUSED_RPN_SOFTMAX), "padded_rois": bool(cfg.PADDED_ROIS),
                       "miopen_find": (not args.no_miopen_benchmark), "miopen_find_db": miopen_db_note,
                       "miopen_searched": miopen_db_grew(miopen_db),
                       "roi_pool_argmax_bytes": leg_meta["argmax_bytes"]},
            "roofline": roofline,
            "scaling_diag": scaling_diag,
            "hot_path": {"gpu_ms_per_step": round(hot_ms, 3),
                         "gpu_images_per_s": round(images_per_step / (hot_ms * 1e-3), 1) if hot_ms > 0 else None,
                         "loss_op_ms_per_step": round(loss_ms, 4), "fused_loss": bool(cfg.get("FUSED_LOSS", True))},
            "final_loss": loss_val,
        }
        if ctx.world_size == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(wl, rois_fixed)
            result["cpu_baseline"] = cb
            gpu_pool_ms = sum(leg[k]["avg_ms"] for k in ("roi_pool_forward", "roi_pool_backward_prepare",
                                                          "roi_pool_backward") if k in leg)
            result["hot_path"]["roi_pool_same_set"] = {
                "R": int(rois_fixed.shape[0]), "gpu_ms": round(gpu_pool_ms, 3),
                "cpu_ms": round(cb["cpu_roi_pool_ms"], 1), "ratio": round(cb["cpu_roi_pool_ms"] / gpu_pool_ms, 1)}
            if hot_ms > 0:
                # NOT like for like: the CPU layers produced and pooled cb["roi_pool_rois"] RoIs, the GPU
                # steps pooled whatever NMS left of this run's proposals (roi_pool_same_set is the
                # same-work figure)
                gpu_rois = [m.get("R") for m in tl.get("roi_pool_forward", {}).get("metas", [])]
                result["hot_path"]["cpu_ms_over_gpu_ms_roi_counts_differ"] = {
                    "ratio": round(cb["cpu_hot_path_ms_per_step"] / hot_ms, 1),
                    "cpu_rois": cb["roi_pool_rois"], "gpu_rois_per_step": [min(gpu_rois), max(gpu_rois)] if gpu_rois else None}
        print(json.dumps(result))
    ctx.shutdown()


if __name__ == "__main__":
    main()
