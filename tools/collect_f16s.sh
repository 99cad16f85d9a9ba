#!/bin/bash
# Evidence for the fp16-storage SPAA modes (configs[1]-f16s, configs[2]-f16s): per-layer tables + bench lines, rocprofv3 kernel stats and
# PMC HBM traffic (separate passes) of the f16s ResNet-18 run.  usage (repo root on the GPU box): bash tools/collect_f16s.sh <tag>
set -eo pipefail
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_f16s
mkdir -p $O
cd $R
python3 bench.py --dtype f16s --no-cpu-baseline --no-modes --profile-out $O/f16s_tapconv_layers.json > $O/f16s_bench.json 2> $O/f16s.log
echo "f16s done"
python3 bench.py --classifier inception_v3 --dtype f16s --steps 10 --no-cpu-baseline --no-modes --profile-out $O/inception_f16s_tapconv_layers.json > $O/inception_f16s_bench.json 2> $O/inc16.log
echo "inception f16s done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_f16s -o kt --output-format csv -- python3 $R/bench.py --dtype f16s --steps 10 --warmup 2 --no-cpu-baseline --no-modes > $O/f16s_under_rocprof.log 2>&1
find $O/stats_f16s -name "*kernel_stats.csv" -exec cp {} $O/f16s_kernel_stats.csv \;
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --dtype f16s --steps 3 --warmup 1 --no-cpu-baseline --no-modes > $O/pmc_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --dtype f16s --steps 3 --warmup 1 --no-cpu-baseline --no-modes > $O/pmc_write.log 2>&1
echo "write pass done"
cd $R
python3 tools/pmc_traffic.py --fetch $O/pmc_fetch --write $O/pmc_write --out $O/f16s_pmc_traffic.json > $O/f16s_pmc_traffic.txt
rm -rf $O/stats_f16s $O/pmc_fetch $O/pmc_write
du -sh $O
