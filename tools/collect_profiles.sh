#!/bin/bash
# Round-end evidence run on the GPU box: rocprofv3 kernel stats, PMC HBM traffic (separate passes), bench line.
# usage (from the repo root on the GPU box): bash tools/collect_profiles.sh <tag>
set -eo pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -A2 -E "FETCH_SIZE|WRITE_SIZE" > $O/counter_units.txt || true
rocprofv3 --kernel-trace --stats -d $O/stats -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-modes > $O/bench_under_rocprof.log 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-modes > $O/pmc_fetch.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-modes > $O/pmc_write.log 2>&1
echo "write pass done"
cd $R
python3 tools/pmc_traffic.py --fetch $O/pmc_fetch --write $O/pmc_write --out $O/pmc_traffic.json | tee $O/pmc_traffic.txt
cp $O/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
python3 bench.py --profile-out $O/tapconv_layers.json > $O/bench.json 2> $O/bench.log
cat $O/bench.json
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/pmc_fetch/*/*.db $O/stats/*/*.db 2>/dev/null || true
du -sh $O
