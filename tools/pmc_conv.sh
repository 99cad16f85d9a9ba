#!/bin/bash
# PMC passes over one conv shape/tile: bash tools/pmc_conv.sh <tag> <prof_conv args...>
set -eo pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES -d $O/a -o a --output-format csv -- python3 $R/tools/${PROF_SCRIPT:-prof_conv.py} "$@" > $O/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/b -o b --output-format csv -- python3 $R/tools/${PROF_SCRIPT:-prof_conv.py} "$@" > $O/b.log 2>&1
cd $R
python3 - $O <<'PY'
import csv,sys,collections,glob
for f in sorted(glob.glob(sys.argv[1]+'/*/*counter_collection.csv')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        import os
        flt = tuple(os.environ.get('PMC_FILTER', 'tapconv,thin,wino,x6p').split(','))
        if any(k in r['Kernel_Name'] for k in flt):
            agg[r['Counter_Name']][r['Kernel_Name'][:60] + '#' + r['Dispatch_Id']]+=float(r['Counter_Value'])
    for c,d in agg.items():
        byk=collections.defaultdict(list)
        for k,v in d.items(): byk[k.split('#')[0]].append(v)
        for k,v in byk.items(): print(f'{c:28s} {sum(v)/len(v):16.0f}  n={len(v)}  {k}')
PY
