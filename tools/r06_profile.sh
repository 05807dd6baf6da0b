#!/bin/bash
# Round-6 evidence run on the GPU box (from the repo root).  Outputs under gpurun_out/r06p/.  Stages (pick with $1, default all): bench prof pmc sq codec small xattn
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06p
mkdir -p $O
cd $R
what=${1:-all}
has() { [ "$what" = all ] || [[ " $what " == *" $1 "* ]]; }
if has bench; then
  timeout 1500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
  timeout 600 python3 bench.py --tier e2e --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b1.json 2> $O/bench_e2e_b1.err; echo "b1 rc=$?"
  timeout 600 python3 bench.py --tier e2e --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b8.json 2> $O/bench_e2e_b8.err; echo "b8 rc=$?"
  timeout 600 python3 bench.py --tier e2e --workload txt2img --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_txt2img_b8.json 2> $O/bench_txt2img_b8.err; echo "txt2img rc=$?"
  timeout 900 python3 bench.py --tier e2e --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b32.json 2> $O/bench_e2e_b32.err; echo "b32 rc=$?"
  timeout 900 python3 bench.py --tier e2e --batch 32 --steps 1 --warmup 1 --image-stages vae+jpeg --no-cpu-baseline > $O/bench_sd21_jpeg_b32.json 2> $O/bench_sd21_jpeg_b32.err; echo "jpeg rc=$?"
  timeout 900 python3 bench.py --tier e2e --batch 16 --steps 1 --warmup 1 --unet sd15 --height 768 --width 768 --no-cpu-baseline > $O/bench_sd15_768_b16.json 2> $O/bench_sd15_768_b16.err; echo "sd15 rc=$?"
  timeout 300 python3 tools/unet_forward_bench.py 32 convs sd15 hw=96 > $O/unet_forward_sd15_768_b32_per_shape.txt 2>&1; head -3 $O/unet_forward_sd15_768_b32_per_shape.txt
  timeout 300 python3 tools/unet_forward_bench.py 128 convs > $O/unet_forward_b128_per_shape.txt 2>&1; head -3 $O/unet_forward_b128_per_shape.txt
  GSW_XATTN_FUSED=0 timeout 300 python3 tools/unet_forward_bench.py 128 > $O/unet_forward_b128_three_launch_cross_attention.txt 2>&1; head -2 $O/unet_forward_b128_three_launch_cross_attention.txt
  timeout 300 python3 tools/xattn_bench.py 128 > $O/xattn_bench.txt 2>&1; timeout 300 python3 tools/xattn_bench.py 8 >> $O/xattn_bench.txt 2>&1; timeout 300 python3 tools/xattn_bench.py 16 sd15 >> $O/xattn_bench.txt 2>&1; cat $O/xattn_bench.txt
fi
cd /tmp && export TMPDIR=/tmp
if has prof; then
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -o e2e -- python3 $R/bench.py --tier e2e --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "rocprof rc=$?"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_codec -o codec -- python3 $R/bench.py --tier codec --no-cpu-baseline > $O/bench_codec_under_rocprof.json 2> $O/bench_codec_under_rocprof.err; echo "rocprof codec rc=$?"
  for d in prof_e2e prof_codec; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
fi
if has small; then
  for rows in 1 16; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rows$rows -o g -- python3 $R/tools/small_rows_profile.py $rows > $O/prof_rows$rows.log 2>&1; echo "graph rows $rows rc=$?"
    f=$(find $O/prof_rows$rows -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/forward_${rows}rows_graph_kernel_stats.csv
  done
fi
if has pmc; then
  for rows in 128 64; do
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${rows}_$c -o p -- python3 $R/tools/unet_forward_bench.py $rows > $O/pmc_${rows}_$c.log 2>&1; echo "pmc $rows $c rc=$?"
    done
    python3 $R/tools/pmc_traffic.py $O/pmc_${rows}_FETCH_SIZE $O/pmc_${rows}_WRITE_SIZE > $O/pmc_unet_forward_b${rows}_traffic.json
  done
  python3 $R/tools/pmc_family.py $O/pmc_unet_forward_b128_traffic.json $O/pmc_unet_forward_b64_traffic.json 64 > $O/e2e_dominant_kernel_pmc.json
fi
if has codec; then
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_codec_$c -o p -- python3 $R/bench.py --tier codec --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_codec_$c.log 2>&1; echo "pmc codec $c rc=$?"
  done
  python3 $R/tools/pmc_traffic.py $O/pmc_codec_FETCH_SIZE $O/pmc_codec_WRITE_SIZE > $O/pmc_codec_traffic.json
fi
if has sq; then
  # SQ counters of the UNet forward at 128 rows with the wide tile on by policy, per kernel instantiation (separate passes: 8 SQ slots)
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout 900 rocprofv3 --pmc $set --output-format csv -d $O/sq_$i -o p -- python3 $R/tools/unet_forward_bench.py 128 > $O/sq_$i.log 2>&1; echo "sq $i rc=$?"
  done
  python3 $R/tools/pmc_sq_summary.py $O/sq_1 $O/sq_2 > $O/unet_forward_b128_sq_pmc_summary.txt 2>&1; head -30 $O/unet_forward_b128_sq_pmc_summary.txt
fi
if has xattn; then
  cd $R; bash tools/xattn_pmc.sh r06x > $O/xattn_pmc.log 2>&1; cp gpurun_out/pmc_r06x_summary.txt $O/xattn_pmc_summary.txt; rm -rf gpurun_out/pmc_r06x_*
  cd $R; bash tools/xattn_pmc.sh r06y pre > $O/xattn_pre_pmc.log 2>&1; cp gpurun_out/pmc_r06y_summary.txt $O/xattn_pre_pmc_summary.txt; rm -rf gpurun_out/pmc_r06y_*
  timeout 600 python3 tools/xattn_ablate.py 128 > $O/xattn_ablate.txt 2>&1; tail -30 $O/xattn_ablate.txt
fi
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O
python3 - <<PY
import json
def show(f):
    try:
        d = json.load(open('$O/' + f + '.json')); r = d.get('roofline', {})
        print(f, round(d['value'], 3), round(d['ms_per_step'], 1), d.get('lossless'), r.get('achieved'), r.get('dense_tflops'), r.get('conv3x3_tflops'), d.get('fallbacks_off_the_hand_written_path'), d.get('board'))
    except Exception as e:
        print(f, 'n/a', e)
for f in ('bench_default', 'bench_e2e_b1', 'bench_e2e_b8', 'bench_txt2img_b8', 'bench_e2e_b32', 'bench_sd21_jpeg_b32', 'bench_sd15_768_b16'):
    show(f)
PY
