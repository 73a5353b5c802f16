#!/bin/bash
# round 3, session 33: embedded complex solvers on half vectors -- regression test, complex suites, timing impact
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s33
O=gpurun_out/s33
timeout -k 10 900 python -m pytest tests/test_gpu_complex.py tests/test_gpu_algorithms.py tests/test_gpu_traces.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python tools/native_vs_embedded.py 16 64 2>&1 | grep -v amdgpu | tee $O/nve.log
timeout -k 10 300 python tools/bench_configs.py ctdvp:32:128 ctdvp:24:512 2>&1 | grep ctdvp | tee $O/ctdvp.log
