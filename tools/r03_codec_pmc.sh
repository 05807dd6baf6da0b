#!/bin/bash
# Codec-tier evidence on the GPU box (from the repo root): the tier's bench line and the two PMC passes (separate runs) of the same command.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
timeout 600 python3 bench.py --tier codec > $O/bench_codec.json 2> $O/bench_codec.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o p -- python3 $R/bench.py --tier codec --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
python3 $R/tools/pmc_traffic.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_codec_traffic.json
cd $R
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
cat $O/pmc_codec_traffic.json | head -30
