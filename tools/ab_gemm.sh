#!/bin/bash
# Same-device A/B of the encoder's split GEMM classes: product library vs the ablation library with a MPREID_GEMM_DBG
# variant (e.g. 64 = operand panels aliased onto L2-resident ones), alternating processes on ONE device in ONE job.
#   bash tools/ab_gemm.sh <dbg> [rounds]
R=${GRAFT_REPO_ROOT:-$(pwd)}
DBG=${1:-64}
N=${2:-2}
ABL=$R/tools/ablation_lib/libmpreid_hip_abl.so
export MPREID_ALLOW_ABLATION=1   # (mpreid/_lib.py refuses an ablation build without it)
for i in $(seq 1 $N); do
  echo "== product"; python3 $R/tools/gemm_bench.py --only split --reps 20 --rounds 3
  echo "== ablation lib, MPREID_GEMM_DBG=0"; MPREID_LIB=$ABL MPREID_GEMM_DBG=0 python3 $R/tools/gemm_bench.py --only split --reps 20 --rounds 3
  echo "== ablation lib, MPREID_GEMM_DBG=$DBG"; MPREID_LIB=$ABL MPREID_GEMM_DBG=$DBG python3 $R/tools/gemm_bench.py --only split --reps 20 --rounds 3
done
