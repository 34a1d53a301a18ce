"""The claim behind EXPERIMENTS.md's closing note on the NMS walk: greedy NMS (cpu_nms.pyx:17-68) has exactly one fixed
point -- kept(i) <=> no kept j < i with ovr(i, j) >= thresh -- and evaluating that rule for all undecided boxes at once
reaches it, i.e. the keep list, in a few rounds.  CPU only: the oracle's NMS against the model of tools/probes."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import np_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("nms_rounds_model", os.path.join(ROOT, "tools", "probes", "nms_rounds_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("thresh", [0.3, 0.5, 0.7])
def test_greedy_nms_is_the_fixed_point_of_its_rule(thresh):
    m = _model()
    rs = np.random.RandomState(int(thresh * 100))
    for n, spread in ((1, 50.0), (64, 80.0), (300, 120.0), (700, 400.0)):
        c = rs.uniform(0, spread, size=(n, 2))
        wh = np.exp(rs.normal(3.5, 0.5, size=(n, 2)))
        boxes = np.hstack((c - wh / 2, c + wh / 2)).astype(np.float32)
        if n > 10:
            boxes[5] = boxes[4]                                  # an exact duplicate
        scores = (np.arange(n, 0, -1, dtype=np.float32) / (n + 1))[:, None]
        want = [int(v) for v in O.nms(np.hstack((boxes, scores)), thresh)]
        status, rounds, _ = m.fixed_point(m.neighbours(boxes, thresh))
        assert [int(v) for v in np.nonzero(status == 1)[0]] == want, (n, thresh)
        assert not (status == 0).any() and rounds <= n + 1
