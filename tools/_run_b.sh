mkdir -p gpurun_out/r2h
python -m pytest tests/test_gpu_rerank.py tests/test_gpu_scale.py -m gpu -x -q > gpurun_out/r2h/pytest.log 2>&1
tail -3 gpurun_out/r2h/pytest.log
for a in "20000 4000 768" "100000 20000 768" "93820 11659 1280" "19281 3368 1280"; do python tools/rerank_bench.py $a 2>&1 | grep "'n'"; done | tee gpurun_out/r2h/rr.log
MPREID_RERANK_NO_OVERLAP=1 python tools/rerank_bench.py 20000 4000 768 2>&1 | grep "'n'"
