#!/bin/bash
# Round-5 job 19: straight-line dense-row epilogue of the wide tile with a ten-deep residual queue: parity, cycle stamps, per-shape table.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_gpu_mm_production.py tests/test_gpu_gemm.py tests/test_gpu_lnfold.py tests/test_gpu_unet_fused.py -q -x > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.txt
for a in "524288 320 2560 1 1 0" "131072 640 1280 0 1 0" "524288 1280 320 0 1 1" "32768 5120 1280 0 1 1" "524288 320 640 0 1 0" "524288 320 320 0 1 1" "131072 2560 640 0 1 1"; do
  timeout 120 tools/ubench/bin/mm_trace_wide $a >> $O/mm_trace_wide.txt 2>&1
done
cat $O/mm_trace_wide.txt
timeout 300 python3 tools/unet_forward_bench.py 128 convs > $O/unet_forward_b128_per_shape.txt 2>&1; head -40 $O/unet_forward_b128_per_shape.txt | cut -c1-190
GSW_MM_WIDE_PMIN_RES=5 timeout 300 python3 tools/unet_forward_bench.py 128 convs > $O/unet_forward_b128_per_shape_res_from5.txt 2>&1; grep -E "B=128|plain\+res" $O/unet_forward_b128_per_shape_res_from5.txt | cut -c1-190
