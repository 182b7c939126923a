"""GPU: the plain-C consumer of the C-ABI (examples/mirl_demo.c, built by
rltime_amd/csrc/build.sh with gcc — no Python, no torch in the process) creates
a shard, ingests, samples, gathers, updates priorities and verifies the gathered
frames itself."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_demo_runs_clean():
    exe = os.path.join(ROOT, "rltime_amd", "csrc", "mirl_demo")
    assert os.path.isfile(exe), "build it with rltime_amd/csrc/build.sh"
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    assert p.stdout.strip().endswith("OK") and "mismatches=0" in p.stdout
