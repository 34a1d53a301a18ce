"""The randomized parity tools of tools/ (each compares the HIP path with the oracle on random inputs: shapes, sizes and
corner cases no fixture holds) run as short tests: a dozen cases each with a seed of their own.  The long runs of the same
tools are recorded in profiles/r04_*_fuzz.log."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("tool,cases", [("nms_fuzz.py", 40), ("roi_pool_fuzz.py", 14), ("layers_fuzz.py", 14),
                                        ("proposal_fuzz.py", 8), ("roi_blocks_fuzz.py", 6), ("image_fuzz.py", 10), ("loss_fuzz.py", 8), ("mil_fuzz.py", 30), ("post_detect_fuzz.py", 20), ("sampler_fuzz.py", 10)])
def test_random_cases_against_the_oracle(tool, cases):
    # a child process per tool (they parse their own command line); one GPU process at a time
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "--cases", str(cases), "--seed", "424242"],
                       cwd=ROOT, capture_output=True, text=True, timeout=580)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0 and "mismatches 0" in r.stdout, tail
