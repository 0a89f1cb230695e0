cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
rm -f gpurun_out/r06e/streams.log
for rep in 1 2; do
for st in 2 3 4 1; do
  python bench.py --steps 4 --warmup 1 --streams $st --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('streams', $st, j['value'], j['ms_per_step'])" >> gpurun_out/r06e/streams.log
done
done
cat gpurun_out/r06e/streams.log
