mkdir -p gpurun_out/r2g
(time python -m pytest tests -m gpu -x -q --durations=12) > gpurun_out/r2g/pytest.log 2>&1
tail -25 gpurun_out/r2g/pytest.log
python bench.py > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err
tail -c 600 gpurun_out/r2g/bench.json
