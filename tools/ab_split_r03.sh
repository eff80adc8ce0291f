mkdir -p gpurun_out/r03
timeout -k 10 300 python -m pytest tests/test_headline_gpu.py -m gpu -x -q -s -k "forward_vs_reference" > gpurun_out/r03/gputests_f.log 2>&1; grep "rel-L2" gpurun_out/r03/gputests_f.log
run() { SEVA_SPLIT_PRECISION=$1 SEVA_ATTN_SPLIT=$2 timeout -k 10 200 python bench.py --no-other-configs --no-cpu-baseline --no-vae --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$2', round(d['ms_per_step'],2), d['roofline']['classes_ms'])" >> gpurun_out/r03/ab_split.log; }
for r in 1 2; do
  run none 2; run stem,head 2; run stem,head,skip_deep 2; run stem,head,skip 2; run stem,head,skip_deep 3; run stem,head,skip_deep 4
done
cat gpurun_out/r03/ab_split.log
