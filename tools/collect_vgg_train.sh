#!/bin/bash
# Evidence for configs[4]'s per-GPU shard (VGG-16 + PerC-AL loop body, fp16 storage) and for the training step:
# per-layer table + bench line + rocprofv3 kernel stats of the VGG run, rocprofv3 kernel stats of tools/lab/train_time.py.
# usage (repo root on the GPU box): bash tools/collect_vgg_train.sh <tag>
set -eo pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_vgg_train
mkdir -p $O
cd $R
python3 bench.py --classifier vgg16 --attack perc_al --dtype f16s --steps 10 --no-cpu-baseline --no-modes --profile-out $O/vgg16_percal_f16s_tapconv_layers.json > $O/vgg16_percal_f16s_bench.json 2> $O/vgg.log
echo "vgg table done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats_vgg -o kt --output-format csv -- python3 $R/bench.py --classifier vgg16 --attack perc_al --dtype f16s --steps 10 --warmup 2 --no-cpu-baseline --no-modes > $O/vgg_under_rocprof.log 2>&1
find $O/stats_vgg -name "*kernel_stats.csv" -exec cp {} $O/vgg16_percal_f16s_kernel_stats.csv \;
echo "vgg rocprof done"
cd $R
rocprofv3 --kernel-trace --stats -d $O/stats_train -o kt --output-format csv -- python3 $R/tools/lab/train_time.py > $O/train_time.log 2>&1
find $O/stats_train -name "*kernel_stats.csv" -exec cp {} $O/train_step_kernel_stats.csv \;
rm -rf $O/stats_vgg $O/stats_train
cat $O/train_time.log | tail -2
