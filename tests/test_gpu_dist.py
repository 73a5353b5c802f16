"""World-size-2 check of the DEVICE-side sharding plumbing on a single-GPU box: two rank processes share GPU 0 and the
collective is staged through the host over gloo (RCCL refuses two ranks on one device), so everything around the
collective -- row-block extraction, mpsk_dAC with Dlo = D / P, re-interleave, the sharded sweep -- is the code the
multi-GPU bench runs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_host_staged():
    port = str(29600 + os.getpid() % 300)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "dist_gpu_check.py"), str(r), "2", port],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o[-2000:]}"
        assert "OK" in o
