#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes over tools/dac_only.py -> per-launch traffic json on stdout
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/tools/dac_only.py > $O/f.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/tools/dac_only.py > $O/w.log 2>&1 || exit 2
cd $R && python tools/pmc_traffic.py $O/f $O/w > $O/traffic.json
python -c "
import json; d=json.load(open('$O/traffic.json'))
for k,v in d.items():
    if k.startswith('dac_gemm') and not k.endswith('detail'): print(k, round(v/1e6,1), 'MB/launch')"
