#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r06
# Writes everything under gpurun_out/profiles_<tag>/; the summaries are then copied into profiles/ (tracked).
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# every bench.py run leaves bench_extras.json in the current directory (/tmp here): keep the ones that are cited
keep_extras() { cp /tmp/bench_extras.json $OUT/$1 2>/dev/null; }
# 1. the bench line EXACTLY as the driver runs it (N = 1: --gpus 1 --steps 20 --warmup 5), and the other workloads
python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/bench.err
keep_extras ${TAG}_bench_extras.json
python3 $R/bench.py --encoder-precision fp16 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_fp16_encoder.json 2>> $OUT/bench.err
python3 $R/bench.py --workload synth --rerank --steps 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_synth_rerank.json 2>> $OUT/bench.err
python3 $R/bench.py --workload synth --dist-mode split3 --steps 3 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_synth_split3.json 2>> $OUT/bench.err
python3 $R/bench.py --workload synth --rerank --rerank-algo split3 --dist-mode split3 --steps 2 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_synth_rerank_split3.json 2>> $OUT/bench.err
python3 $R/bench.py --workload msmt17 --rerank --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_msmt17_rerank.json 2>> $OUT/bench.err
python3 $R/bench.py --rerank --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_market_rerank.json 2>> $OUT/bench.err
keep_extras ${TAG}_bench_market_rerank_extras.json
# 2. kernel trace + stats of the same bench command (--streams 1 is honoured by every leg, also the host-loader ones of
#    `extras`: no two kernels of the process overlap, so a kernel's average duration here is its launch duration)
#    --no-extras: only the headline step's launches are in the trace (the extras run the same kernels at other shapes -- the
#    reference-loop leg encodes 64 images per call -- and rocprofv3 --stats averages per kernel NAME)
#    Same --steps / --warmup as the driver's run, so that `Calls` are the driver run's launch counts.
#    MPREID_EVAL_D2H=behind: the matrix's D2H copy is queued BEHIND eval_rank_kernel in this trace only.  The product overlaps them
#    (faster: tools/evalrank_bench.py), but rocprofv3 serialises the two queues and then books the blit kernels the ranking
#    kernel waited for into ITS interval (round 5's 5.4 ms average); ordered this way the trace shows the kernel's own time.
export MPREID_EVAL_D2H=behind
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_bench -o b -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-extras > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
unset MPREID_EVAL_D2H
keep_extras ${TAG}_bench_under_rocprof_extras.json
cp $OUT/kt_bench/b_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
# 3. re-ranking alone (N = 20 000): kernel stats and the FETCH / WRITE passes (separate runs)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_rr -o rr -- python3 $R/tools/rerank_bench.py 20000 4000 768 > $OUT/rr.log 2>&1
cp $OUT/kt_rr/rr_kernel_stats.csv $OUT/${TAG}_rerank_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_rr_fetch -o rr -- python3 $R/tools/rerank_bench.py 20000 4000 768 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_rr_write -o rr -- python3 $R/tools/rerank_bench.py 20000 4000 768 > /dev/null 2>&1
# 4. the 20k x 20k x 768 feat-GEMM (stored, fp16 one pass): SQ counters, FETCH, WRITE
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_euclid -o e -- python3 $R/tools/distgemm_bench.py 20000 20000 768 > $OUT/distgemm.log 2>&1
cp $OUT/kt_euclid/e_kernel_stats.csv $OUT/${TAG}_featgemm_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_e_sq -o e -- python3 $R/tools/distgemm_bench.py 20000 20000 768 f16 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_e_fetch -o e -- python3 $R/tools/distgemm_bench.py 20000 20000 768 f16 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_e_write -o e -- python3 $R/tools/distgemm_bench.py 20000 20000 768 f16 > /dev/null 2>&1
# 4b. the same matrix from TWO tensors (the evaluator's shape: every tile computed, 256 x 256 one-workgroup-per-CU kernel)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_euclid2 -o e -- python3 $R/tools/distgemm_bench.py 20000 20000 768 f16 full > $OUT/distgemm_two_tensors.log 2>&1
cp $OUT/kt_euclid2/e_kernel_stats.csv $OUT/${TAG}_featgemm_two_tensors_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_e2_sq -o e -- python3 $R/tools/distgemm_bench.py 20000 20000 768 f16 full > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_e2_fetch -o e -- python3 $R/tools/distgemm_bench.py 20000 20000 768 f16 full > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_e2_write -o e -- python3 $R/tools/distgemm_bench.py 20000 20000 768 f16 full > /dev/null 2>&1
# 4c. eval_rank_kernel alone / beside the D2H copy, compute() in both orders
python3 $R/tools/evalrank_bench.py > $OUT/${TAG}_evalrank.log 2>&1
# 5. encoder GEMM classes (fp16 and split-precision): FETCH / WRITE passes over the micro-benchmark (tools/pmc_traffic.py reduces them)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_g_fetch -o g -- python3 $R/tools/gemm_bench.py --reps 3 --rounds 2 --only qkv,out,fc1,fc2,sqkv,sout,sfc1,sfc2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_g_write -o g -- python3 $R/tools/gemm_bench.py --reps 3 --rounds 2 --only qkv,out,fc1,fc2,sqkv,sout,sfc1,sfc2 > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $OUT/pmc_g_fetch $OUT/pmc_g_write $OUT/${TAG}_gemm_pmc_traffic.json > /dev/null 2>&1
# 5b. matrix-pipe counters of the same classes (the dominant kernel of the headline step is split_fc_bias_quickgelu): one SQ / GRBM pass
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_g_mfma -o g -- python3 $R/tools/gemm_bench.py --reps 3 --rounds 2 --only qkv,out,fc1,fc2,sqkv,sout,sfc1,sfc2 > /dev/null 2>&1
python3 $R/tools/pmc_mfma.py $OUT/pmc_g_mfma $OUT/${TAG}_gemm_pmc_mfma.json > $OUT/pmc_mfma.log 2>&1
python3 $R/tools/pmc_summary.py $OUT $TAG > $OUT/${TAG}_pmc_summary.json 2> $OUT/pmc_summary.err
ls -la $OUT | head -40
# 6. same-device A/B of this round's one encoder change (tile order of the split FC2: MPREID_TUNE=gemm_walk=0 is round 5's order),
#    alternating processes, the headline step
ab() { env "$@" python3 $R/bench.py --gpus 1 --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-live-traffic 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$LABEL', j['value'], 'img/s', j['ms_per_step'], 'ms/step')"; }
for i in 1 2 3; do
  LABEL="round 6 (gemm_walk auto)" ab MPREID_AB=product >> $OUT/${TAG}_ab_gemm_walk.log
  LABEL="round 5 order (gemm_walk=0)" ab MPREID_TUNE=gemm_walk=0 >> $OUT/${TAG}_ab_gemm_walk.log
done
cat $OUT/${TAG}_ab_gemm_walk.log
