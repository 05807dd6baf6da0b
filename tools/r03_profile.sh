#!/bin/bash
# Round-3 evidence run on the GPU box (from the repo root): headline bench, rocprofv3 kernel stats of the same command, PMC traffic passes of the
# UNet forward at the two row counts of a step.  Outputs under gpurun_out/r03p/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -o e2e -- python3 $R/bench.py --tier e2e --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "rocprof rc=$?"
for rows in 128 64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${rows}_$c -o p -- python3 $R/tools/unet_forward_bench.py $rows > $O/pmc_${rows}_$c.log 2>&1; echo "pmc $rows $c rc=$?"
  done
  python3 $R/tools/pmc_traffic.py $O/pmc_${rows}_FETCH_SIZE $O/pmc_${rows}_WRITE_SIZE > $O/pmc_unet_forward_b${rows}_traffic.json
done
cd $R
find $O -name "*kernel_stats.csv" | head -2
# keep the merge-back small: drop the raw traces
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
du -sh $O
