#!/bin/bash
# Round-5 job 14: polynomial GELU (parity of everything that applies it), start stagger of the engine's workgroups (A/B per shape and end to end).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05n
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_lnfold.py tests/test_gpu_mm_production.py tests/test_gpu_pf.py tests/test_gpu_small.py tests/test_gpu_splitk.py tests/test_gpu_unet_fused.py tests/test_gpu_graph.py tests/test_gpu_fullsize.py -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for s in 0 2 4 8; do
  GSW_MM_STAGGER=$s timeout 300 python3 tools/unet_forward_bench.py 128 convs > $O/unet_forward_b128_stagger$s.txt 2>&1; head -3 $O/unet_forward_b128_stagger$s.txt | tail -2
done
GSW_MM_STAGGER=4 GSW_MM_STAGGER_TBPS=3.5 timeout 300 python3 tools/unet_forward_bench.py 128 convs > $O/unet_forward_b128_stagger4_slow.txt 2>&1; head -3 $O/unet_forward_b128_stagger4_slow.txt | tail -2
for s in 0 4; do
  GSW_MM_STAGGER=$s timeout 300 python3 tools/unet_forward_bench.py 64 > $O/unet_forward_b64_stagger$s.txt 2>&1; tail -1 $O/unet_forward_b64_stagger$s.txt
done
for s in 0 4 0 4; do
  GSW_MM_STAGGER=$s timeout 900 python3 bench.py --tier e2e --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b64_stagger$s.json 2> $O/bench_e2e_b64_stagger$s.err; echo "b64 stagger $s rc=$?"
  python3 -c "
import json
d=json.load(open('$O/bench_e2e_b64_stagger$s.json')); r=d['roofline']; print('stagger $s', d['value'], r['achieved'], r['conv3x3_tflops'], r['dense_tflops'], d['board']['sclk_mhz_mean'], d['board']['power_w_mean'])"
done
