# A/B runs of the orbit headline with option sets given one per line on stdin (tools only)
mkdir -p gpurun_out/r3
while read -r o; do
  timeout -k 10 120 python bench.py --cpu-frames 0 --steps 120 --warmup 10 --no-extras $o > gpurun_out/r3/s.json || exit 1
  python -c "
import json,sys; d=json.load(open('gpurun_out/r3/s.json')); print('$o', round(d['ms_per_step'],4), 'pairs', d.get('tile_pairs'), 'iters', d.get('render_iters'))"
done
