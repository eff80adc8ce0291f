#!/bin/bash
# PMC passes over one kernel of a small driver script (one counter group per pass; kernel-trace only, never combined with
# runtime traces).   usage: tools/pmc_kernel.sh <driver.py> <kernel-name substring[|substring...]> <tag>   -> gpurun_out/pmc_<tag>/summary.txt
set -e
drv=$1; pat=$2; tag=$3
out=$PWD/gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o p$i -- python3 $drv > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/p$i.log; continue; }
  find $out/p$i -name "*.db" -delete; find $out/p$i -name "*kernel_trace.csv" -delete
done
PAT="$pat" OUT="$out" python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["OUT"]
pats = os.environ["PAT"].split("|")  # several kernels of one driver: "patA|patB" -> one block per pattern
rows = []
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
with open(out + "/summary.txt", "w") as fo:
    for pat in pats:
        tot = collections.defaultdict(list)
        for r in rows:
            if pat in r["Kernel_Name"]:
                tot[r["Counter_Name"]].append(float(r["Counter_Value"]))
        fo.write(f"kernel pattern: {pat}\n"); print("kernel pattern:", pat)
        for k in sorted(tot):
            v = tot[k]
            line = f"{k:28s} launches {len(v)}  mean {sum(v)/len(v):.4g}"
            print(line); fo.write(line + "\n")
PY
find $out -name "*counter_collection.csv" -delete
