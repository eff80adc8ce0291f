# A/B of two builds of the library on one box: bench.py with the production library vs SEVA_HIP_LIB=$1; interleaved rounds.
# usage: bash tools/ab_lib.sh build_ab/libseva_hip_old.so [rounds]
mkdir -p gpurun_out/r04
log=gpurun_out/r04/ab_lib.log; rm -f $log
run() { SEVA_HIP_LIB=$2 timeout -k 10 200 python bench.py --no-other-configs --no-cpu-baseline --no-vae --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],2), d['roofline']['classes_ms'])" >> $log; tail -1 $log; }
for r in $(seq 1 ${2:-2}); do run other $PWD/$1; run production ""; done
