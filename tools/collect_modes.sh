#!/bin/bash
# Evidence for the non-headline modes: per-layer tables + bench lines (f16s, Inception-v3 f32 / f16s) and rocprofv3 kernel stats of
# the f16s and Inception f32 runs.  usage (repo root on the GPU box): bash tools/collect_modes.sh <tag>
set -eo pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_modes
mkdir -p $O
cd $R
python3 bench.py --dtype f16s --no-cpu-baseline --no-modes --profile-out $O/f16s_tapconv_layers.json > $O/f16s_bench.json 2> $O/f16s.log
echo "f16s done"
python3 bench.py --classifier inception_v3 --steps 10 --no-cpu-baseline --no-modes --profile-out $O/inception_f32_tapconv_layers.json > $O/inception_f32_bench.json 2> $O/inc32.log
echo "inception f32 done"
python3 bench.py --classifier inception_v3 --dtype f16s --steps 10 --no-cpu-baseline --no-modes --profile-out $O/inception_f16s_tapconv_layers.json > $O/inception_f16s_bench.json 2> $O/inc16.log
echo "inception f16s done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_f16s -o kt --output-format csv -- python3 $R/bench.py --dtype f16s --steps 10 --warmup 2 --no-cpu-baseline --no-modes > $O/f16s_under_rocprof.log 2>&1
find $O/stats_f16s -name "*kernel_stats.csv" -exec cp {} $O/f16s_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats -d $O/stats_inc -o kt --output-format csv -- python3 $R/bench.py --classifier inception_v3 --steps 5 --warmup 2 --no-cpu-baseline --no-modes > $O/inc_under_rocprof.log 2>&1
find $O/stats_inc -name "*kernel_stats.csv" -exec cp {} $O/inception_f32_kernel_stats.csv \;
rm -rf $O/stats_f16s $O/stats_inc
echo "rocprof passes done"
