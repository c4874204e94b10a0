#!/bin/bash
# A/B of library builds on the bench workload: tools/ab.sh <lib.so | ""> ...   ("" = the in-tree libgsx.so)
for v in "$@"; do
  GSX_LIB=$v timeout -k 10 150 python bench.py --cpu-frames 0 --steps 100 > gpurun_out/ab_one.json 2> gpurun_out/ab_one.err || { echo "FAILED $v"; tail -5 gpurun_out/ab_one.err; exit 1; }
  python - "$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_one.json"))
s = d["stage_ms"]
print(f"{sys.argv[1] or 'in-tree':28s} ms/frame {d['ms_per_step']:.4f} pairs {d['tile_pairs']} sort {s['keygen_ms']+s['sort_ms']:.3f} bin {s['bin_ms']:.3f} raster {s['raster_ms']:.3f} scatter_frac {d['roofline']['frac']:.3f}", flush=True)
PY
done
