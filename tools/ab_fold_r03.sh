# A/B on one box: the ResBlock's 1x1 skip conv as its own GEMM (0) vs folded into the second 3x3 conv (1); two interleaved rounds.
mkdir -p gpurun_out/r03
rm -f gpurun_out/r03/ab_fold.log
run() { SEVA_FOLD_SKIP=$1 timeout -k 10 200 python bench.py --no-other-configs --no-cpu-baseline --no-vae --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold_skip=$1', round(d['ms_per_step'],2), d['roofline']['classes_ms'])" >> gpurun_out/r03/ab_fold.log; tail -1 gpurun_out/r03/ab_fold.log; }
for r in 1 2; do run 0; run 1; done
SEVA_FOLD_SKIP=1 timeout -k 10 300 python -m pytest tests/test_headline_gpu.py -m gpu -x -q -s -k "forward_vs_reference" 2>&1 | grep "rel-L2" | tee -a gpurun_out/r03/ab_fold.log
