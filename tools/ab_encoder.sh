#!/bin/bash
# Same-device A/B of the headline step (processor.do_inference over the Market-1501 shape): the product library against
# another build of it (default: tools/ablation_lib/libmpreid_hip_base.so = the current tree with the previous round's
# vit.hip / gemm_f16.hip / common.h, built by
#   MPREID_CSRC=<scratch csrc> MPREID_BUILD_TAG=base python mp-reid_amd/mpreid/build.py),
# alternating processes on ONE device in ONE job: devices of the pool differ by 2-3 % in the clock they hold.
#   bash tools/ab_encoder.sh [other.so] [rounds] [steps]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=${1:-$R/tools/ablation_lib/libmpreid_hip_base.so}
N=${2:-3}
STEPS=${3:-5}
run() {  # label, env...
  local label=$1; shift
  env "$@" python3 $R/bench.py --gpus 1 --steps $STEPS --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null |
    python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$label', j['value'], 'img/s', j['ms_per_step'], 'ms/step | dominant', r['kernel'], r['avg_launch_ms'], 'ms')"
  python3 -c "
import json
d=json.load(open('bench_extras.json'))
print('   ', ' | '.join(f\"{c['kernel'].split('<')[1][:-1]} {c['N']}x{c['K']} {c['avg_ms']}\" for c in d['gemm_classes'][:4]), '| att', d['other_encoder_kernels'][0]['avg_ms'], '| ln', d['other_encoder_kernels'][1]['avg_ms'])"
}
for i in $(seq 1 $N); do
  run "product " MPREID_AB=product
  run "other   " MPREID_LIB=$OTHER
done
