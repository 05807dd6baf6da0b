#!/bin/bash
# Round-5 job 5: whole GPU suite on the new kernels (wide tile + residual touches, attention tail), thresholds again, attention shapes, codec on one / two streams.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
for cfg in "0 8 64" "7 8 64" "7 8 40" "7 8 16" "7 5 40"; do
  set -- $cfg
  echo "== GSW_MM_WIDE=$1 PMIN=$2 PMIN_PF=$3" >> $O/unet_forward_b128.txt
  GSW_MM_WIDE=$1 GSW_MM_WIDE_PMIN=$2 GSW_MM_WIDE_PMIN_PF=$3 timeout 300 python3 tools/unet_forward_bench.py 128 convs >> $O/unet_forward_b128.txt 2>&1
done
grep -E "==|SD 2.1" $O/unet_forward_b128.txt
timeout 600 python3 tools/attn_shapes_bench.py 32 sd15 > $O/attn_shapes_b32.txt 2>&1; cat $O/attn_shapes_b32.txt
for st in 1 2; do timeout 300 python3 bench.py --tier codec --codec-streams $st --no-cpu-baseline > $O/bench_codec_streams$st.json 2> $O/bench_codec_streams$st.err; done
python3 -c "
import json
for st in (1,2):
    d=json.load(open('$O/bench_codec_streams%d.json'%st)); r=d['roofline']; print('codec streams',st, d['value'], d['ms_per_step'], r['kernels'], r.get('step_GBps'))
"
