#!/bin/bash
# Round-3 evidence for the LOD cut of cfg 3 (6 M splats, 8.48 M octree nodes): kernel stats and roofline of the three-launch
# cut and of round 2's launch-per-level form on the same box, the compaction on its own, the pipelined frame time of both.
#   tools/lod_evidence_r3.sh   (on the GPU box; writes gpurun_out/r3/lod_*)
mkdir -p gpurun_out/r3
timeout -k 10 300 bash tools/lod_profile.sh r3 > gpurun_out/r3/lod_profile_new.txt 2>&1 || exit 1
cp gpurun_out/r3_cfg3_lod_roofline.json gpurun_out/r3/lod_roofline_new.json
cp gpurun_out/r3_cfg3_kernel_stats.txt gpurun_out/r3/lod_kernel_stats_new.txt
GSX_LOD_EXTRA_OPTS="lod_walk=2" timeout -k 10 300 bash tools/lod_profile.sh r3old > gpurun_out/r3/lod_profile_old.txt 2>&1 || exit 1
cp gpurun_out/r3old_cfg3_lod_roofline.json gpurun_out/r3/lod_roofline_old.json
cp gpurun_out/r3old_cfg3_kernel_stats.txt gpurun_out/r3/lod_kernel_stats_old.txt
timeout -k 10 200 tools/compactbench 8484260 30 > gpurun_out/r3/compactbench.txt 2>&1 || exit 1
for i in 1 2 3; do
  timeout -k 10 200 python tools/configs_table.py cfg3 2>/dev/null || exit 1
  GSX_OPTS="lod_walk=2" timeout -k 10 200 python tools/configs_table.py cfg3 2>/dev/null || exit 1
done > gpurun_out/r3/lod_cfg3_frames.jsonl
grep -h "cut_us\|median\|us_per_cut" gpurun_out/r3/lod_roofline_new.json gpurun_out/r3/lod_roofline_old.json
cut -c1-100 gpurun_out/r3/compactbench.txt | head -4
python - <<'PY'
import json
for l in open("gpurun_out/r3/lod_cfg3_frames.jsonl"):
    d = json.loads(l); print(d["opts"] if "opts" in d else "", d["ms_per_frame"])
PY
