#!/bin/bash
# Round-5 job 2: first run of the 256 x 320 tile (MT = 8) of the matmul engine: parity at the production shapes, then the same launches on both tiles.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_mm_production.py -q -x -k "WIDE and not unet and not vae" > $O/pytest_wide.txt 2>&1; echo "pytest wide rc=$?"; tail -5 $O/pytest_wide.txt
for w in 0 2; do
  echo "== GSW_MM_WIDE=$w" >> $O/power_probe_mm.txt
  GSW_MM_WIDE=$w timeout 300 python3 tools/power_probe.py mm >> $O/power_probe_mm.txt 2>&1
done
cat $O/power_probe_mm.txt
for w in 0 2; do
  echo "== GSW_MM_WIDE=$w" >> $O/unet_forward_b128.txt
  GSW_MM_WIDE=$w timeout 300 python3 tools/unet_forward_bench.py 128 convs >> $O/unet_forward_b128.txt 2>&1
done
grep -E "SD 2.1|conv3x3|up2x" $O/unet_forward_b128.txt | head -80
