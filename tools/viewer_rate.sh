#!/bin/bash
# Frame rate of the viewer front end (gaussian-splatting_amd/gsx_view) at cfg 2's size: 6 M synthetic splats, 1920x1080, a 300-frame
# mouse drag; frames presented into host memory (no files).   tools/viewer_rate.sh <outdir>
set -e
out=${1:-gpurun_out/viewer}
mkdir -p $out
python3 - "$out" <<'PY'
import importlib.util, os, sys
spec = importlib.util.spec_from_file_location("synth", "gaussian-splatting_amd/synth.py")
synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)
n = int(os.environ.get("GSX_VIEW_SPLATS", 6_000_000))
synth.clustered_cloud(n, seed=6).tofile(os.path.join(sys.argv[1], "scene.splat"))
with open(os.path.join(sys.argv[1], "drag.txt"), "w") as f:
    f.write("frame 20\nmove 400 500\nbutton 0 1\n")
    for k in range(300):
        f.write(f"move {400 + 3 * k} {500 - (k % 40)}\nframe\n")
    f.write("button 0 0\nscroll 40\nframe 40\n")
PY
for mesh in base octree; do
    timeout -k 10 600 gaussian-splatting_amd/gsx_view $out/scene.splat --mesh $mesh --size 1920x1080 --events $out/drag.txt --stats 60 2> $out/rate_$mesh.txt
    tail -4 $out/rate_$mesh.txt
    timeout -k 10 600 gaussian-splatting_amd/gsx_view $out/scene.splat --mesh $mesh --size 1920x1080 --events $out/drag.txt --stats 60 --pipelined 2> $out/rate_${mesh}_pipelined.txt
    tail -4 $out/rate_${mesh}_pipelined.txt
done
rm -f $out/scene.splat
