#!/bin/bash
# compile one csrc file and print its register / scratch statistics
cd /root/repo/spaa_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $1.hip -o /tmp/$1.o -save-temps=obj 2>&1 | grep -E "error" ; grep -E "^\s+\.(name|vgpr_count|private_segment_fixed_size|vgpr_spill_count):" /tmp/$1-hip-amdgcn-amd-amdhsa-gfx950.s
