#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of one conv shape: bash tools/pmc_fetch.sh <tag> <prof_conv args...>
set -eo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmcf_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 $R/tools/prof_conv.py "$@" > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 $R/tools/prof_conv.py "$@" > $O/w.log 2>&1
cd $R
python3 tools/pmc_traffic.py --fetch $O/f --write $O/w --out $O/traffic.json | grep tapconv
