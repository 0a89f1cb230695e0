#!/bin/bash
# ragged row counts on the XCD-owned walk (MPREID_TUNE=gemm_ragged=0 is round 5's rule): the GEMM classes at the patch embedding's
# M (65024 = 254 tile rows) and at a 485-image encode group's (62720 = 245 tile rows), then the headline step, alternating
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/ragged
rm -f gpurun_out/ragged/ragged.log
for r in 1 0; do
  for m in 65024 62720; do
    echo "== gemm_ragged=$r M=$m" >> gpurun_out/ragged/ragged.log
    MPREID_TUNE=gemm_ragged=$r python tools/gemm_bench.py --m $m --reps 20 --rounds 3 --only sqkv,sout,sfc1,sfc2 2>/dev/null >> gpurun_out/ragged/ragged.log
  done
done
for i in 1 2 3; do
  for r in 1 0; do
    MPREID_TUNE=gemm_ragged=$r python bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline gemm_ragged=$r', j['value'], 'img/s', j['ms_per_step'], 'ms/step')" >> gpurun_out/ragged/ragged.log
  done
done
cat gpurun_out/ragged/ragged.log
