#!/bin/bash
# Round-5 job 3: the wide tile on every epilogue it has (convolutions, dense rows, GEGLU, +- LayerNorm fold): parity, then A/B by epilogue mask.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_mm_production.py tests/test_gpu_lnfold.py tests/test_gpu_gemm.py -q -x -k "not unet and not vae" > $O/pytest_wide.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_wide.txt
for w in 0 7; do
  echo "== GSW_MM_WIDE=$w" >> $O/power_probe.txt
  GSW_MM_WIDE=$w timeout 300 python3 tools/power_probe.py mm >> $O/power_probe.txt 2>&1
  GSW_MM_WIDE=$w timeout 300 python3 tools/power_probe.py geglu >> $O/power_probe.txt 2>&1
done
grep -E "==|conv3x3|dense|geglu" $O/power_probe.txt
for w in 0 2 7; do
  echo "== GSW_MM_WIDE=$w" >> $O/unet_forward_b128.txt
  GSW_MM_WIDE=$w timeout 300 python3 tools/unet_forward_bench.py 128 convs >> $O/unet_forward_b128.txt 2>&1
done
grep -E "==|SD 2.1" $O/unet_forward_b128.txt
