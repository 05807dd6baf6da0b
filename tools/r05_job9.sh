#!/bin/bash
# Round-5 job 9: the rotated step sequence of the wide tile (DMA in the odd tails, first barrier of a tile without a vmcnt wait): parity, then A/B.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05i
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_mm_production.py tests/test_gpu_lnfold.py tests/test_gpu_gn_colstats.py tests/test_gpu_gemm.py -q -x -k "not vae" > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for cfg in "0 5 64" "7 5 64" "7 5 40" "7 5 24"; do
  set -- $cfg
  echo "== GSW_MM_WIDE=$1 PMIN=$2 PMIN_PF=$3" >> $O/unet_forward_b128.txt
  GSW_MM_WIDE=$1 GSW_MM_WIDE_PMIN=$2 GSW_MM_WIDE_PMIN_PF=$3 timeout 300 python3 tools/unet_forward_bench.py 128 convs >> $O/unet_forward_b128.txt 2>&1
done
grep -E "==|SD 2.1" $O/unet_forward_b128.txt
timeout 1200 python3 bench.py --tier e2e --no-cpu-baseline > $O/bench_e2e_b64_wide.json 2> $O/bench_e2e_b64_wide.err; echo "wide rc=$?"
GSW_MM_WIDE=0 timeout 1200 python3 bench.py --tier e2e --no-cpu-baseline > $O/bench_e2e_b64_narrow.json 2> $O/bench_e2e_b64_narrow.err; echo "narrow rc=$?"
python3 -c "
import json
for f in ('bench_e2e_b64_wide','bench_e2e_b64_narrow'):
    d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, round(d['value'],3), round(d['ms_per_step']), d['lossless'], 'fam', round(r['achieved']), 'dense', round(r.get('dense_tflops',0)), 'conv', round(r.get('conv3x3_tflops',0)), d['board']['sclk_mhz_mean'], d['board']['power_w_mean'], d['fallbacks_off_the_hand_written_path'])
"
