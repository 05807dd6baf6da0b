#!/bin/bash
# Round-5 job 13: GroupNorm-only convolutions without border zeroing (parity), the wide tile's cycle stamps, the rounds-based wide rule on the SD 1.5 / 768 shape.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05m
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_gn_colstats.py tests/test_gpu_unet_fused.py tests/test_gpu_graph.py tests/test_gpu_fullsize.py tests/test_gpu_mm_production.py tests/test_gpu_small.py -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for a in "524288 320 2560 1 1 0" "131072 640 5120 1 1 0" "32768 1280 10240 1 1 0" "131072 640 1280 0 1 0" "524288 1280 320 0 1 1" "32768 5120 1280 0 1 1" "524288 320 640 0 1 0"; do
  timeout 120 tools/ubench/bin/mm_trace_wide $a >> $O/mm_trace_wide.txt 2>&1
done
cat $O/mm_trace_wide.txt
timeout 900 python3 bench.py --tier e2e --batch 16 --steps 1 --warmup 1 --unet sd15 --height 768 --width 768 --no-cpu-baseline > $O/bench_sd15_768_b16.json 2> $O/bench_sd15_768_b16.err; echo "sd15 rc=$?"
timeout 600 python3 bench.py --tier e2e --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b8.json 2> $O/bench_e2e_b8.err; echo "b8 rc=$?"
timeout 600 python3 bench.py --tier e2e --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b1.json 2> $O/bench_e2e_b1.err; echo "b1 rc=$?"
timeout 1200 python3 bench.py --tier e2e --no-cpu-baseline > $O/bench_e2e_b64.json 2> $O/bench_e2e_b64.err; echo "b64 rc=$?"
python3 -c "
import json
for f in ('bench_sd15_768_b16','bench_e2e_b8','bench_e2e_b1','bench_e2e_b64'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['lossless'], d['fallbacks_off_the_hand_written_path'], d['roofline'].get('achieved'))
"
