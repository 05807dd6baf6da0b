#!/bin/bash
# Round-5 job 4: the default bench with the wide tile on (policy from job 3), and the same with GSW_MM_WIDE=0 on the same box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
timeout 1200 python3 bench.py --tier e2e --no-cpu-baseline > $O/bench_e2e_b64_wide.json 2> $O/bench_e2e_b64_wide.err; echo "wide rc=$?"
GSW_MM_WIDE=0 timeout 1200 python3 bench.py --tier e2e --no-cpu-baseline > $O/bench_e2e_b64_narrow.json 2> $O/bench_e2e_b64_narrow.err; echo "narrow rc=$?"
timeout 900 python3 bench.py --tier e2e --batch 16 --steps 1 --warmup 1 --unet sd15 --height 768 --width 768 --no-cpu-baseline > $O/bench_sd15_768_b16.json 2> $O/bench_sd15_768_b16.err; echo "sd15 rc=$?"
python3 -c "
import json
for f in ('bench_e2e_b64_wide','bench_e2e_b64_narrow','bench_sd15_768_b16'):
    d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, round(d['value'],3), round(d['ms_per_step']), d['lossless'], 'fam', round(r['achieved']), 'dense', round(r.get('dense_tflops',0)), 'conv', round(r.get('conv3x3_tflops',0)), d['board']['sclk_mhz_mean'], d['board']['power_w_mean'], d['fallbacks_off_the_hand_written_path'])
"
