#!/bin/bash
# A/B of the frame loop's schedule knobs on cfg 2 (bench.py, no extras, no CPU baseline): ms per step, pairs, iterations per setting
for opts in "" "--opt tile_caps=0" "--opt chunk_first=1073741824" "--opt chunk_first=1048576" "--opt chunk_first=524288" "--opt chunk_first=131072" "--opt chunk_first=65536 --opt chunk_growth=16" "--opt chunk_first=1073741824 --opt tile_caps=0"; do
  r=$(timeout -k 10 120 python bench.py --cpu-frames 0 --no-extras $opts 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), d['tile_pairs'], d['render_iters'])")
  echo "$opts => $r"
done
