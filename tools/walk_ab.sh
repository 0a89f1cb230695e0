cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
rm -f gpurun_out/r06c/grid.log
for rep in 1 2; do
  echo "== 256 CUs, M=65536 (walk 3)" >> gpurun_out/r06c/grid.log
  MPREID_TUNE=gemm_walk=3 python tools/gemm_bench.py --reps 20 --rounds 3 --only sqkv,sout,sfc1,sfc2 2>/dev/null >> gpurun_out/r06c/grid.log
  echo "== 224 CUs, M=57344 (walk 3)" >> gpurun_out/r06c/grid.log
  MPREID_TUNE=gemm_walk=3,gemm_grid=224 python tools/gemm_bench.py --m 57344 --reps 20 --rounds 3 --only sqkv,sout,sfc1,sfc2 2>/dev/null >> gpurun_out/r06c/grid.log
  echo "== 192 CUs, M=49152 (walk 3)" >> gpurun_out/r06c/grid.log
  MPREID_TUNE=gemm_walk=3,gemm_grid=192 python tools/gemm_bench.py --m 49152 --reps 20 --rounds 3 --only sqkv,sout,sfc1,sfc2 2>/dev/null >> gpurun_out/r06c/grid.log
done
cat gpurun_out/r06c/grid.log
