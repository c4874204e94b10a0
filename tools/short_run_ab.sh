# what the sampled sorts cost the driver's 20-step run (tools only)
mkdir -p gpurun_out/r3
for se in 4 8 1024; do for rep in 1 2; do
GSX_BENCH_SAMPLE_EVERY=$se timeout -k 10 120 python bench.py --cpu-frames 0 --steps 20 --warmup 5 --no-extras > gpurun_out/r3/s.json || exit 1
python -c "
import json; d=json.load(open('gpurun_out/r3/s.json')); print('sample_every', $se, round(d['ms_per_step'],4), d['roofline']['launches_timed'], d.get('sort_only_ms'))"
done; done
