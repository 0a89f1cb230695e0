#!/bin/bash
# encode group size of do_inference's pipeline (bench.py --group): 508 images = one sweep of 256 x 256 row tiles over the 256 CUs per
# GEMM launch, 1016 = two sweeps, 1524 = three.  Same-device alternation of the headline step.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/group_ab
rm -f gpurun_out/group_ab/group.log
for rep in 1 2; do
  for g in 508 1016 1524 254; do
    python bench.py --group $g --steps 4 --warmup 1 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group', $g, j['value'], 'img/s', j['ms_per_step'], 'ms/step')" >> gpurun_out/group_ab/group.log
  done
done
cat gpurun_out/group_ab/group.log
