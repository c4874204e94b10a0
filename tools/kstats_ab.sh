# per-kernel durations (serialized: one stream, one frame at a time) for two option sets (tools only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
for o in "$1" "$2"; do
rm -rf gpurun_out/prof_ab
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ab -o ab -- python3 bench.py --cpu-frames 0 --steps 60 --warmup 10 --no-extras --in-order --opt async_sort=0 --opt frames_in_flight=1 $o > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
echo "== options: $o"
python tools/kstats.py gpurun_out/prof_ab | head -12
done
rm -rf gpurun_out/prof_ab
