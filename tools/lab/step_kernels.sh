#!/bin/bash
# Per-step kernel list: rocprofv3 kernel stats of bench.py at 5 and at 15 timed steps; tools/lab/step_kernels.py takes the difference.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/stepk
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/a -o kt --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-modes > $O/a.log 2>&1
echo "pass a done"
rocprofv3 --kernel-trace --stats -d $O/b -o kt --output-format csv -- python3 $R/bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-modes > $O/b.log 2>&1
echo "pass b done"
find $O/a -name "*kernel_stats.csv" -exec cp {} $O/a_stats.csv \;
find $O/b -name "*kernel_stats.csv" -exec cp {} $O/b_stats.csv \;
rm -rf $O/a $O/b
