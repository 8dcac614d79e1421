#!/bin/bash
# usage (GPU box): tools/prof_all.sh <round tag, e.g. r2>  — every profile the round's numbers are quoted from, summarised ON the
# box (the raw traces exceed what gpurun merges back) into gpurun_out/<tag>_*.{md,json,csv}:
#   1. rocprofv3 --kernel-trace --stats of the default bench command            -> <tag>_bench_train_bf16.md (+ kernel stats csv, bench line)
#   2. separate --pmc FETCH_SIZE / WRITE_SIZE passes (kernel trace only)        -> <tag>_pmc_traffic_train_bf16.{json,md}
#   3. --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass                       -> <tag>_pmc_mfma_busy_train_bf16.md
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_$TAG $O/pmc_$TAG $O/pmcg_mfma $O/osd_tuner_$TAG.json
# one set of tuner choices for every pass: the first run tunes and writes the cache, the others load it — the same kernels in the
# same order, so the counter passes can be joined with the kernel trace by (kernel, occurrence index)
export OSD_TUNER_CACHE=$O/osd_tuner_$TAG.json      # under the run's own scratch directory, not a fixed /tmp name
bash $R/tools/prof_bench.sh $TAG --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/summarize_profile.py $O/prof_$TAG $O/${TAG}_bench_train_bf16.md > /dev/null 2>&1
cp $O/prof_$TAG/run_kernel_stats.csv $O/${TAG}_bench_train_bf16_kernel_stats.csv 2>/dev/null
grep '^{"metric"' $O/prof_$TAG/bench.log | tail -1 > $O/${TAG}_bench_train_bf16_line_under_profiler.json
# a trace of the counter passes' own command (4 steps) for the join
rocprofv3 --kernel-trace --output-format csv -d $O/prof_${TAG}_short -o run -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rm -rf $O/prof_$TAG
bash $R/tools/prof_pmc.sh $TAG --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/summarize_pmc.py $O/pmc_$TAG $O/${TAG}_pmc_traffic_train_bf16 > /dev/null 2>&1
rm -rf $O/pmc_$TAG
bash $R/tools/prof_pmc_generic.sh mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" bench.py --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/summarize_mfma.py $O/pmcg_mfma $O/${TAG}_pmc_mfma_busy_train_bf16.md $(find $O/prof_${TAG}_short -name '*kernel_trace.csv' | head -1) > /dev/null 2>&1
rm -rf $O/pmcg_mfma $O/prof_${TAG}_short
ls -la $O | grep ${TAG}_
