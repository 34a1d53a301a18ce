"""The C ABI from plain C (tests/abi_c/abi_harness.c: gcc, the HIP runtime's C API for device memory,
no Python / torch / C++ on the calling side).  CPU: the harness compiles against include/wssdl_bus_hip.h
and links against the built library.  -m gpu: its results equal the oracle's."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi_c", "abi_harness.c")


def build(tmpdir):
    from wssdl_bus_amd import build as B
    lib = B.build(verbose=False)
    exe = os.path.join(str(tmpdir), "abi_harness")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-O1", "-Wall", "-Werror", "-I" + os.path.join(rocm, "include"), "-I" + os.path.join(ROOT, "include"),
           SRC, "-o", exe, "-L" + os.path.dirname(lib), "-lwssdl_bus_hip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath," + os.path.join(rocm, "lib")]
    subprocess.check_call(cmd)
    return exe


def test_c_harness_compiles_and_links(tmp_path):
    exe = build(tmp_path)
    out = subprocess.check_output(["nm", "-u", exe]).decode()
    for sym in ("wssdl_bbox_overlaps", "wssdl_nms", "wssdl_roi_pool_forward", "wssdl_generate_anchors_host"):
        assert sym in out                                   # bound through the header's prototypes, resolved at load


@pytest.mark.gpu
def test_c_harness_results_equal_the_oracle(tmp_path):
    from conftest import load_golden
    from oracle import c_oracle
    exe = build(tmp_path)
    rs = np.random.RandomState(17)
    N, K = 37, 5
    boxes = np.sort(rs.uniform(0, 600, (N, 4)), axis=1)[:, [0, 1, 2, 3]].astype(np.float64)
    boxes = np.stack([np.minimum(boxes[:, 0], boxes[:, 2]), np.minimum(boxes[:, 1], boxes[:, 3]),
                      np.maximum(boxes[:, 0], boxes[:, 2]), np.maximum(boxes[:, 1], boxes[:, 3])], 1)
    query = boxes[rs.choice(N, K, replace=False)] + rs.uniform(-20, 20, (K, 4))
    n_det = 300
    xy = rs.uniform(0, 400, (n_det, 2))
    wh = rs.uniform(20, 200, (n_det, 2))
    dets = np.concatenate([xy, xy + wh, rs.permutation(n_det).reshape(-1, 1) / float(n_det)], 1).astype(np.float32)
    fN, fH, fW, fC, R = 2, 12, 17, 8, 9
    feat = np.maximum(rs.normal(size=(fN, fH, fW, fC)), 0).astype(np.float32)
    rois = np.concatenate([rs.randint(0, fN, (R, 1)), np.sort(rs.uniform(0, 16 * fW - 1, (R, 2)), 1)[:, :1],
                           np.sort(rs.uniform(0, 16 * fH - 1, (R, 2)), 1)[:, :1], np.zeros((R, 2))], 1).astype(np.float32)
    rois[:, 3] = np.minimum(rois[:, 1] + rs.uniform(10, 150, R), 16 * fW - 1)
    rois[:, 4] = np.minimum(rois[:, 2] + rs.uniform(10, 150, R), 16 * fH - 1)

    def fmt(a):
        return " ".join(repr(float(v)) for v in np.asarray(a).ravel())
    text = "anchors\n"
    text += "iou %d %d\n%s\n%s\n" % (N, K, fmt(boxes), fmt(query))
    text += "nms %d 0.7\n%s\n" % (n_det, fmt(dets))
    text += "pool %d %d %d %d %d\n%s\n%s\n" % (fN, fH, fW, fC, R, fmt(feat), fmt(rois))
    p = subprocess.run([exe], input=text.encode(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = p.stdout.decode().split("\n")
    i = 0
    assert lines[i] == "anchors 9"
    anchors = np.array([[float(v) for v in l.split()] for l in lines[i + 1:i + 10]])
    assert np.array_equal(anchors, load_golden("anchors")["a_8_16_32"])
    i += 10
    assert lines[i] == "iou %d %d" % (N, K)
    iou = np.array([[float(v) for v in l.split()] for l in lines[i + 1:i + 1 + N]])
    assert np.array_equal(iou, c_oracle.bbox_overlaps(boxes, query))          # f64, bit for bit (%.17g round-trips)
    i += 1 + N
    n_keep = int(lines[i].split()[1])
    keep = [int(v) for v in lines[i + 1].split()]
    want = list(c_oracle.cpu_nms(dets, 0.7))
    assert n_keep == len(want) and keep == [int(v) for v in want]
    i += 2
    nt = int(lines[i].split()[1])
    vals = np.array([l.split() for l in lines[i + 1:i + 1 + nt]])
    top = vals[:, 0].astype(np.float32).reshape(R, 7, 7, fC)
    arg = vals[:, 1].astype(np.int32).reshape(R, 7, 7, fC)
    et, ea = c_oracle.roi_pool_forward(feat, rois, 7, 7, 1.0 / 16, "cuda")
    assert np.array_equal(top, et) and np.array_equal(arg, ea)
    assert lines[i + 1 + nt].startswith("done wssdl_bus_hip")
