"""Stage and import the reference's *Python* hot-path layers in THIS container only
(TEST INFRASTRUCTURE: used by tests/golden/make_golden.py to generate golden
vectors and by tests that cross-check the oracle when /root/reference exists).

Nothing staged here is kept: the scratch tree lives under a temp dir outside
the repo, and only input/output *data* (``tests/golden/*.npz``) is committed.

Mechanical steps applied to the scratch copy:
  * ``lib2to3`` (print statements, xrange, implicit relative import);
  * ``/`` -> ``//`` for the py2 integer division
    ``cfg.TRAIN.BATCH_SIZE / num_images`` (proposal_target_layer_tf_bus.py:57,135);
  * aliases ``np.float/np.int/np.bool`` (removed from NumPy >= 1.24);
  * an attribute-dict module named ``easydict`` (the package is not installed;
    it carries no arithmetic);
  * empty ``fast_rcnn/__init__.py`` (the real one imports TensorFlow modules);
  * ``utils/cython_bbox.so`` etc. are symlinks to ``oracle/_ref/*.so`` built by
    ``oracle/build_ref.py`` from the reference's own .pyx files.
"""
import atexit
import importlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("WSSDL_REFERENCE", "/root/reference")
LIB = os.path.join(REF, "code", "lib")

_PY_FILES = [
    "rpn_msr/generate_anchors.py",
    "rpn_msr/anchor_target_layer_tf_bus.py",
    "rpn_msr/proposal_layer_tf_bus.py",
    "rpn_msr/proposal_target_layer_tf_bus.py",
    "fast_rcnn/config.py",
    "fast_rcnn/bbox_transform.py",
    "fast_rcnn/nms_wrapper.py",
    "utils/timer.py",
]

_EASYDICT = '''
class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)
    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        super().__setitem__(k, v)
    __setitem__ = __setattr__
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)
    def has_key(self, k):
        return k in self
'''

_staged = None


def reference_present():
    return os.path.isdir(LIB)


def stage():
    """Create the scratch tree, put it on sys.path, return its path."""
    global _staged
    if _staged is not None:
        return _staged
    if not reference_present():
        raise RuntimeError("reference tree %s not present" % REF)
    from . import build_ref
    build_ref.build(verbose=False)
    root = tempfile.mkdtemp(prefix="wssdl_ref_scratch_")
    atexit.register(shutil.rmtree, root, True)
    for rel in _PY_FILES:
        dst = os.path.join(root, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copy(os.path.join(LIB, rel), dst)
    for d in ("rpn_msr", "fast_rcnn", "utils", "nms"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
        open(os.path.join(root, d, "__init__.py"), "w").close()
    subprocess.check_call(
        [sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", root],
        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    p = os.path.join(root, "rpn_msr", "proposal_target_layer_tf_bus.py")
    txt = open(p).read()
    assert txt.count("cfg.TRAIN.BATCH_SIZE / num_images") == 2
    txt = txt.replace("cfg.TRAIN.BATCH_SIZE / num_images",
                      "cfg.TRAIN.BATCH_SIZE // num_images")
    open(p, "w").write(txt)
    with open(os.path.join(root, "easydict.py"), "w") as f:
        f.write(_EASYDICT)
    ref_so = os.path.join(HERE, "_ref")
    os.symlink(os.path.join(ref_so, "cython_bbox.so"),
               os.path.join(root, "utils", "cython_bbox.so"))
    os.symlink(os.path.join(ref_so, "cython_bbox_ui.so"),
               os.path.join(root, "utils", "cython_bbox_ui.so"))
    os.symlink(os.path.join(ref_so, "cpu_nms.so"),
               os.path.join(root, "nms", "cpu_nms.so"))
    os.symlink(os.path.join(ref_so, "cython_nms.so"),
               os.path.join(root, "utils", "cython_nms.so"))
    for name, typ in (("float", float), ("int", int), ("bool", bool)):
        if not hasattr(np, name):
            setattr(np, name, typ)
    sys.path.insert(0, root)
    _staged = root
    return root


class _Ref(object):
    """Namespace with the imported reference callables."""


def load():
    stage()
    import warnings
    warnings.filterwarnings("ignore", category=DeprecationWarning)
    r = _Ref()
    ga = importlib.import_module("rpn_msr.generate_anchors")
    at = importlib.import_module("rpn_msr.anchor_target_layer_tf_bus")
    pl = importlib.import_module("rpn_msr.proposal_layer_tf_bus")
    pt = importlib.import_module("rpn_msr.proposal_target_layer_tf_bus")
    bt = importlib.import_module("fast_rcnn.bbox_transform")
    cf = importlib.import_module("fast_rcnn.config")
    nw = importlib.import_module("fast_rcnn.nms_wrapper")
    cb = importlib.import_module("utils.cython_bbox")
    cu = importlib.import_module("utils.cython_bbox_ui")
    cn = importlib.import_module("nms.cpu_nms")
    un = importlib.import_module("utils.cython_nms")      # the test path's NMS (fast_rcnn/test_bus.py:10)
    r.cfg = cf.cfg
    r.generate_anchors = ga.generate_anchors
    r.anchor_target_layer = at.anchor_target_layer
    r.anchor_target_layer_ws = at.anchor_target_layer_ws
    r.anchor_target_layer_joint = at.anchor_target_layer_joint
    r.proposal_layer = pl.proposal_layer
    r.proposal_target_layer = pt.proposal_target_layer
    r.proposal_target_layer_joint = pt.proposal_target_layer_joint
    r.bbox_transform = bt.bbox_transform
    r.bbox_transform_inv = bt.bbox_transform_inv
    r.clip_boxes = bt.clip_boxes
    r.nms = nw.nms
    r.bbox_overlaps = cb.bbox_overlaps
    r.bbox_overlaps_ui = cu.bbox_overlaps_ui
    r.cpu_nms = cn.cpu_nms
    r.utils_nms = un.nms
    r.utils_nms_new = un.nms_new
    r.modules = dict(at=at, pl=pl, pt=pt)
    return r
