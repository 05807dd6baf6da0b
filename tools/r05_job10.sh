#!/bin/bash
# Round-5 job 10: rotated step sequence with the DMA spread (3 pieces in the odd tail + 6 in the even phase; 9 in front of an epilogue): parity + A/B on one box.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_mm_production.py tests/test_gpu_lnfold.py tests/test_gpu_gn_colstats.py -q -x -k "not vae" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for w in 0 7; do
  echo "== GSW_MM_WIDE=$w" >> $O/power_probe.txt
  GSW_MM_WIDE=$w timeout 300 python3 tools/power_probe.py mm >> $O/power_probe.txt 2>&1
  GSW_MM_WIDE=$w timeout 300 python3 tools/power_probe.py geglu >> $O/power_probe.txt 2>&1
done
grep -E "==|conv3x3|dense|geglu" $O/power_probe.txt
for w in 0 7; do
  echo "== GSW_MM_WIDE=$w" >> $O/unet_forward_b128.txt
  GSW_MM_WIDE=$w timeout 300 python3 tools/unet_forward_bench.py 128 convs >> $O/unet_forward_b128.txt 2>&1
done
grep -E "==|SD 2.1" $O/unet_forward_b128.txt
timeout 1200 python3 bench.py --tier e2e --no-cpu-baseline > $O/bench_e2e_b64_wide.json 2> $O/bench_e2e_b64_wide.err; echo "wide rc=$?"
GSW_MM_WIDE=0 timeout 1200 python3 bench.py --tier e2e --no-cpu-baseline > $O/bench_e2e_b64_narrow.json 2> $O/bench_e2e_b64_narrow.err; echo "narrow rc=$?"
python3 -c "
import json
for f in ('bench_e2e_b64_wide','bench_e2e_b64_narrow'):
    d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, round(d['value'],3), round(d['ms_per_step']), d['lossless'], 'fam', round(r['achieved']), 'dense', round(r.get('dense_tflops',0)), 'conv', round(r.get('conv3x3_tflops',0)), d['board']['sclk_mhz_mean'], d['board']['power_w_mean'], d['fallbacks_off_the_hand_written_path'])
"
