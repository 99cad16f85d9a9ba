#!/bin/bash
# A/B on one box: ReLU gates of the VGG-16 / Inception-v3 bodies as byte masks (SPAA_BODY_MASKS=1, default) against the activation
# itself as the gate (0); VGG-16's first layer on the two-half smallcin kernel against the register-staged tile
# (SPAA_DEFAULT_DISABLE=smallcin64).  usage (repo root on the GPU box): bash tools/lab/body_masks_ab.sh
set -eo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/body_masks_ab
mkdir -p $O
cd $R
line() { python3 -c "
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d['value'], 'it/s', d['ms_per_step'], 'ms')" $1 "$2"; }
for m in 0 1; do
  SPAA_BODY_MASKS=$m python3 bench.py --classifier vgg16 --attack perc_al --dtype f16s --steps 10 --no-cpu-baseline --no-modes > $O/vgg_f16s_m$m.json 2> $O/vgg_f16s_m$m.log
  line $O/vgg_f16s_m$m.json "vgg16 perc_al f16s masks=$m"
done
SPAA_DEFAULT_DISABLE=smallcin64 python3 bench.py --classifier vgg16 --attack perc_al --dtype f16s --steps 10 --no-cpu-baseline --no-modes > $O/vgg_f16s_nosc.json 2> $O/vgg_f16s_nosc.log
line $O/vgg_f16s_nosc.json "vgg16 perc_al f16s masks=1, first layer on the register-staged tile"
for d in f32 f16s; do for m in 0 1; do
  SPAA_BODY_MASKS=$m python3 bench.py --classifier inception_v3 --dtype $d --steps 10 --no-cpu-baseline --no-modes > $O/inc_${d}_m$m.json 2> $O/inc_${d}_m$m.log
  line $O/inc_${d}_m$m.json "inception_v3 $d masks=$m"
done; done
for m in 0 1; do
  SPAA_BODY_MASKS=$m python3 bench.py --classifier vgg16 --dtype f32 --steps 5 --no-cpu-baseline --no-modes > $O/vgg_f32_m$m.json 2> $O/vgg_f32_m$m.log
  line $O/vgg_f32_m$m.json "vgg16 spaa f32 masks=$m"
done
