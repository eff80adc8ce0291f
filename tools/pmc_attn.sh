#!/bin/bash
# PMC passes over the attention kernel alone (one counter group per pass, kernel-trace only).
set -e
out=$PWD/gpurun_out/pmc_attn
mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o p$i -- python3 tools/kattn_one.py > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; continue; }
  find $out/p$i -name "*.db" -delete; find $out/p$i -name "*kernel_trace.csv" -delete
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.getcwd(), "gpurun_out", "pmc_attn")
tot = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_kernel" in r["Kernel_Name"]:
            tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(tot):
        v = tot[k]
        line = f"{k:28s} launches {len(v)}  mean {sum(v)/len(v):.4g}"
        print(line); fo.write(line + "\n")
PY
