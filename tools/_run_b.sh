mkdir -p gpurun_out/r2e
python -m pytest tests/test_gpu_rerank.py -m gpu -x -q > gpurun_out/r2e/pytest.log 2>&1
tail -3 gpurun_out/r2e/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2e/kt -o rr -- python3 $GRAFT_REPO_ROOT/tools/rerank_bench.py 20000 4000 768 > $GRAFT_REPO_ROOT/gpurun_out/r2e/rr.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/r2e/kt
