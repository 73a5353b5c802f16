#!/bin/bash
# Round-end validation on the GPU box: tests, PMC traffic passes, bench (+ rocprofv3 kernel stats), secondary configs.
# Everything lands under gpurun_out/final/ ; copy what is to be judged into profiles/.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -5 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/dac_only.py > $O/pmc_fetch.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/dac_only.py > $O/pmc_write.log 2>&1 || exit 2
cd $R
python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/pmc_traffic.json && cp $O/pmc_traffic.json profiles/r01_pmc_traffic.json
timeout -k 10 900 python bench.py > $O/bench_n1.log 2>&1 || { tail -5 $O/bench_n1.log; exit 3; }
tail -1 $O/bench_n1.log > $O/bench_n1.json; cut -c1-400 $O/bench_n1.json
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --no-cpu-baseline > $O/prof_bench.log 2>&1 || exit 4
cd $R
timeout -k 10 300 python tools/bench_site.py > $O/site_breakdown.log 2>&1
timeout -k 10 600 python tools/bench_configs.py c2 c3 c4 tsvd c4sweep c4sweep:16:1024 ctdvp > $O/other_configs.log 2>&1
for a in "4096 graded pre" "2048 graded pre" "1024 graded pre"; do timeout -k 10 120 python tools/svd_only.py $a 2>&1 | grep "^tsvd" >> $O/other_configs.log; done
timeout -k 10 200 python tools/bench_qp.py 1024 > $O/bench_qp.log 2>&1
MPSK_BENCH_PROF=1 timeout -k 10 300 python tools/bench_dac.py 1024,2,5 2048,2,5 1024,4,6 256,3,5 512,2,3 > $O/bench_dac.log 2>&1
echo done
