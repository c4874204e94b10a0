#!/bin/bash
# cfg 3 (6 M splats, 8.48 M octree nodes, 120-frame orbit) under rocprofv3 --kernel-trace --stats: per-kernel time of the
# LOD cut beside the sort and the frame.   tools/lod_profile.sh <tag>
set -e
tag=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_${tag}_cfg3
# (one stream, one frame at a time: the per-kernel durations are then the kernels' own)
export GSX_OPTS="async_sort=0 frames_in_flight=1 $GSX_LOD_EXTRA_OPTS"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_cfg3 -o cfg3 -- python3 tools/configs_table.py cfg3 > gpurun_out/${tag}_cfg3.json 2> gpurun_out/${tag}_cfg3.err
unset GSX_OPTS
python tools/kstats.py gpurun_out/prof_${tag}_cfg3 > gpurun_out/${tag}_cfg3_kernel_stats.txt
cat gpurun_out/${tag}_cfg3.json gpurun_out/${tag}_cfg3_kernel_stats.txt
timeout -k 10 200 python tools/lod_roofline.py gpurun_out/prof_${tag}_cfg3 > gpurun_out/${tag}_cfg3_lod_roofline.json && cat gpurun_out/${tag}_cfg3_lod_roofline.json
