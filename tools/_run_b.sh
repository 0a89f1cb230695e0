python -m pytest tests/test_gpu_rerank.py -m gpu -x -q 2>&1 | tail -2
for a in "20000 4000 768" "19281 3368 1280" "100000 20000 768"; do python tools/rerank_bench.py $a 2>&1 | grep "'n'"; done
