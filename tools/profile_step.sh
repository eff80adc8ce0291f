#!/bin/bash
# One eager (no hipGraph) bench step under rocprofv3: kernel-trace stats + two PMC passes (FETCH_SIZE, WRITE_SIZE).
# Run on the GPU box from the repo root; results land in gpurun_out/prof_<tag>/.
set -e
tag=${1:-r04}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp SEVA_HIPGRAPH=0
args="bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vae --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 $args > $out/bench_stats.json 2> $out/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 $args > /dev/null 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python3 $args > /dev/null 2> $out/write.err
python3 tools/traffic_from_pmc.py $out/fetch $out/write $out/traffic.json > /dev/null
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
# keep only the small artefacts
find $out -name "*.db" -delete
find $out -name "*kernel_trace.csv" -delete
find $out -name "*.csv" -size +8M -delete
ls -la $out
