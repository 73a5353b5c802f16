#!/bin/bash
# Round-end validation on the GPU box: tests, smoke, PMC traffic passes, bench (+ rocprofv3 kernel stats), secondary configs.
# Everything lands under gpurun_out/final/ ; the files to be judged are copied into profiles/.
# usage: tools/final_validation.sh [part]   part = all | core | configs
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
PART=${1:-all}
TAG=${TAG:-r03}
if [ "$PART" = all ] || [ "$PART" = core ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest_gpu.log 2>&1 || { tail -5 $O/${TAG}_pytest_gpu.log; exit 1; }
  tail -1 $O/${TAG}_pytest_gpu.log
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
  tail -1 $O/smoke.log
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/dac_only.py > $O/pmc_fetch.log 2>&1 || exit 2
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/dac_only.py > $O/pmc_write.log 2>&1 || exit 2
  timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/tools/dac_only.py > $O/pmc_mfma.log 2>&1
  cd $R
  python tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write > $O/${TAG}_pmc_traffic.json && cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
  for f in $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1); do cp $f $O/${TAG}_pmc_fetch_size_dac_only.csv; done
  for f in $(find $O/pmc_write -name "*counter_collection.csv" | head -1); do cp $f $O/${TAG}_pmc_write_size_dac_only.csv; done
  for f in $(find $O/pmc_mfma -name "*counter_collection.csv" | head -1); do cp $f $O/${TAG}_pmc_mfma_busy_dac_only.csv; done
  timeout -k 10 900 python bench.py > $O/bench_n1.log 2>&1 || { tail -5 $O/bench_n1.log; exit 3; }
  tail -1 $O/bench_n1.log > $O/${TAG}_bench_n1.json; cut -c1-300 $O/${TAG}_bench_n1.json
  cd /tmp
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --no-cpu-baseline --no-tolerance-sweep > $O/prof_bench.log 2>&1 || exit 4
  cd $R
  for f in $(find $O/prof_bench -name "*kernel_stats.csv" | head -1); do cp $f $O/${TAG}_kernel_stats.csv; done
  timeout -k 10 300 python bench.py --steps 1 --warmup 1 --force-shard --no-cpu-baseline --no-tolerance-sweep > $O/${TAG}_bench_force_shard.json 2> $O/force_shard.err
fi
if [ "$PART" = all ] || [ "$PART" = configs ]; then
  timeout -k 10 300 python tools/bench_site.py > $O/${TAG}_site_breakdown.log 2>&1
  timeout -k 10 1100 python tools/bench_configs.py c2 c3 c4 tsvd c4sweep:16:1024 c4sweep:64:1024:3 ctdvp:32:128 ctdvp:24:512 > $O/${TAG}_other_configs.log 2>&1
  timeout -k 10 400 python tools/svd_probe.py 1024,2048,4096 graded6 graded12 uniform > $O/${TAG}_svd_modes.log 2>&1
  timeout -k 10 300 python tools/split_probe.py 3 > $O/${TAG}_split_probe.log 2>&1
  timeout -k 10 200 python tools/bench_cplx.py 128 256 512 1024 > $O/${TAG}_bench_cplx.log 2>&1
  (timeout -k 10 100 python tools/qr_only.py 2048 1024; timeout -k 10 100 python tools/qr_only.py 768 256; timeout -k 10 100 python tools/qr_only.py 4096 4096 3; timeout -k 10 200 python tools/qr_fallback_cost.py) > $O/${TAG}_qr_only.log 2>&1
  MPSK_BENCH_PROF=1 timeout -k 10 300 python tools/bench_dac.py 1024,2,5 2048,2,5 1024,4,6 256,3,5 512,2,3 > $O/${TAG}_bench_dac.log 2>&1
  timeout -k 10 400 python bench.py --L 28 --D 4096 --steps 1 --warmup 1 --no-cpu-baseline --no-tolerance-sweep > $O/${TAG}_bench_D4096_L28.json 2>/dev/null
  cd /tmp
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 $R/tools/bench_configs.py c4sweep:16:1024:1 > $O/prof_c4.log 2>&1
  cd $R
  for f in $(find $O/prof_c4 -name "*kernel_stats.csv" | head -1); do cp $f $O/${TAG}_kernel_stats_c4sweep.csv; done
fi
echo done
