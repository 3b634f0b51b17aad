"""A short fixed-seed run of each randomized fuzz tool (tools/fuzz_*.py: random shapes, encoder inputs, sampler branches and harness calls
against the oracle) - the tools themselves run hundreds of cases (profiles/r05_fuzz_*.txt).  Needs an MI355X."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,cases,seed", [("fuzz_shapes.py", 10, 11), ("fuzz_encoder.py", 24, 12), ("fuzz_sampler.py", 14, 13),
                                             ("fuzz_harness.py", 8, 14), ("fuzz_evaluate.py", 10, 15)])
def test_fuzz_tool_short_run(tool, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(cases), str(seed)], capture_output=True, text=True, timeout=180)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert f"{cases} cases, 0 failures" in r.stdout
