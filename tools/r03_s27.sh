#!/bin/bash
# round 3, session 27: long fused Gram-Schmidt passes: test, then the tolerance-mode legs (bench to-tolerance, VUMPS D = 512)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/s27
O=gpurun_out/s27
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_traces.py tests/test_gpu_algorithms.py -x -q -m gpu -k "orth_step or vectors or vumps or ritz or eigsolve or multilincomb or quasiparticle" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python tools/bench_configs.py c3 2>&1 | grep "c3 " | tee $O/c3.log
timeout -k 10 600 python bench.py --no-cpu-baseline --early-sweeps 0 > $O/bench.log 2>&1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s27/bench.log").read().strip().splitlines()[-1])
print("bench", d["value"], [(r["sweep"], r["ms"], r["matvecs_per_site_mean"]) for r in d["to_tolerance"]["sweeps"]])
PY
