#!/bin/bash
# A/B of gsx_set_option settings on the bench workload: tools/ab_opt.sh "name=v,name=v" ...   ("" = defaults)
for v in "$@"; do
  args=(); IFS=',' read -ra kv <<< "$v"; for o in "${kv[@]}"; do [ -n "$o" ] && args+=(--opt "$o"); done
  timeout -k 10 150 python bench.py --cpu-frames 0 --steps 100 "${args[@]}" > gpurun_out/ab_one.json 2> gpurun_out/ab_one.err || { echo "FAILED $v"; tail -5 gpurun_out/ab_one.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_one.json"))
s = d["stage_ms"]
print(f"{sys.argv[1] or 'defaults':40s} ms/frame {d['ms_per_step']:.4f} pairs {d['tile_pairs']} iters {d['render_iters']} sort {s['keygen_ms']+s['sort_ms']:.3f} bin {s['bin_ms']:.3f} raster {s['raster_ms']:.3f}", flush=True)
PY
done
