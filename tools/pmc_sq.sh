#!/bin/bash
# SQ-level counter passes over tools/dac_only.py (one rocprofv3 --pmc run per counter group) -> per-kernel means per launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcsq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_LDS SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python3 $R/tools/dac_only.py "$@" > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $O/p$i.log; }
done
cd $R && python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('gpurun_out/pmcsq/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('mpsk::', '').replace(' ', '')
        if not k.startswith('dac_gemm') and not k.startswith('gemm_sk_fixup'):
            continue
        a = acc[k][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, v in acc.items():
    print(k)
    for c, (s, n) in sorted(v.items()):
        print(f"   {c:36s} {s / n:16.1f}   (n={n})")
PY
