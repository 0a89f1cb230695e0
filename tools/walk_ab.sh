#!/bin/bash
# Tile order / grid size of the persistent 256 x 256 GEMM, split-precision encoder classes (all forms bit-identical:
# tests/test_gpu_vit.py::test_persistent_gemm_tile_orders_are_bit_identical).  Run through gpurun from the repo root:
#   bash tools/walk_ab.sh            -> gpurun_out/walk_ab/walk.log   (profiles/r06_gemm_walk_ab.log)
#   bash tools/walk_ab.sh grid       -> gpurun_out/walk_ab/grid.log   (the same work per CU on 224 / 192 CUs:
#                                                                      profiles/r06_cu_partition_experiment.log)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/walk_ab
if [ "$1" = "grid" ]; then
  rm -f gpurun_out/walk_ab/grid.log
  for rep in 1 2; do
    for cfg in "256 65536" "224 57344" "192 49152"; do
      set -- $cfg
      echo "== $1 CUs, M=$2 (groups of 4 tile rows)" >> gpurun_out/walk_ab/grid.log
      MPREID_TUNE=gemm_walk=3,gemm_grid=$1 python tools/gemm_bench.py --m $2 --reps 20 --rounds 3 --only sqkv,sout,sfc1,sfc2 2>/dev/null >> gpurun_out/walk_ab/grid.log
    done
  done
  cat gpurun_out/walk_ab/grid.log
else
  rm -f gpurun_out/walk_ab/walk.log
  for rep in 1 2; do
    for w in 0 1 2 3 4; do
      echo "== gemm_walk=$w" >> gpurun_out/walk_ab/walk.log
      MPREID_TUNE=gemm_walk=$w python tools/gemm_bench.py --reps 20 --rounds 3 --only sqkv,sout,sfc1,sfc2 2>/dev/null >> gpurun_out/walk_ab/walk.log
    done
  done
  cat gpurun_out/walk_ab/walk.log
fi
