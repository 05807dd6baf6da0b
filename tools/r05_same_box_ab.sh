#!/bin/bash
# Round-5 job 23: the straight-line dense-row epilogue (global stores, residual queue) and the tile-top fragment reads: parity, then a same-box A/B against the engine of
# commit f48ec25 (tools/ubench/bin/libgswm_old.so = that gswm_mm.hip linked with today's other objects:
#   mkdir -p /tmp/o/a/b /tmp/o/include; for f in gswm_mm.hip gswm_mm.h gswm_mmtypes.h gswm_ablate.inc; do git show f48ec25:a-watermark-for-diffusion-models_amd/csrc/$f > /tmp/o/a/b/$f; done; cp include/gswm.h /tmp/o/include/
#   (cd /tmp/o/a/b && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -c gswm_mm.hip -o /tmp/mm_old.o)
#   cd a-watermark-for-diffusion-models_amd/csrc && hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=gswm.map build/gswm_{kernels,conv,image,attn}.o /tmp/mm_old.o build/gswm_small.o -o ../../tools/ubench/bin/libgswm_old.so ).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05q
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_mm_production.py tests/test_gpu_gemm.py tests/test_gpu_lnfold.py tests/test_gpu_unet_fused.py tests/test_gpu_splitk.py tests/test_gpu_small.py tests/test_gpu_graph.py tests/test_gpu_gn_colstats.py -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
OLD=$R/tools/ubench/bin/libgswm_old.so
for rep in 1 2; do
  GSWM_LIB=$OLD timeout 300 python3 tools/unet_forward_bench.py 128 convs > $O/unet_forward_b128_old_$rep.txt 2>&1; head -3 $O/unet_forward_b128_old_$rep.txt | tail -2
  timeout 300 python3 tools/unet_forward_bench.py 128 convs > $O/unet_forward_b128_new_$rep.txt 2>&1; head -3 $O/unet_forward_b128_new_$rep.txt | tail -2
done
GSWM_LIB=$OLD timeout 300 python3 tools/unet_forward_bench.py 64 > $O/unet_forward_b64_old.txt 2>&1; tail -1 $O/unet_forward_b64_old.txt
timeout 300 python3 tools/unet_forward_bench.py 64 > $O/unet_forward_b64_new.txt 2>&1; tail -1 $O/unet_forward_b64_new.txt
for v in old new old new; do
  if [ $v = old ]; then export GSWM_LIB=$OLD; else unset GSWM_LIB; fi
  timeout 900 python3 bench.py --tier e2e --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b64_$v.json 2> $O/bench_e2e_b64_$v.err; echo "b64 $v rc=$?"
  python3 -c "
import json
d=json.load(open('$O/bench_e2e_b64_$v.json')); r=d['roofline']; print('$v', d['value'], r['achieved'], r['conv3x3_tflops'], r['dense_tflops'], d['board']['sclk_mhz_mean'], d['board']['power_w_mean'])"
done
