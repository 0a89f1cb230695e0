cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/att6 -o a -- python3 $GRAFT_REPO_ROOT/tools/vit_once.py 508 > /dev/null 2>&1
echo "$(grep 'attention_kernelILi10ELi4ELb1' $GRAFT_REPO_ROOT/gpurun_out/att6/a_kernel_stats.csv | cut -d, -f1-4)"
cd $GRAFT_REPO_ROOT; timeout 600 python -m pytest tests/test_gpu_vit.py -x -q 2>&1 | tail -2
