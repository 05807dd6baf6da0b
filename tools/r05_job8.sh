#!/bin/bash
# Round-5 evidence run on the GPU box (from the repo root): attention with the permuted V^T rows, the bench lines of every configuration, rocprofv3 kernel
# statistics of the default and the SD 1.5 / 768 x 768 runs, PMC traffic of the engine family.  Outputs under gpurun_out/r05h/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05s
mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.txt
timeout 600 python3 tools/attn_shapes_bench.py 32 all > $O/attn_shapes_b32.txt 2>&1; cat $O/attn_shapes_b32.txt
timeout 1200 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python3 bench.py --tier e2e --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b1.json 2> $O/bench_e2e_b1.err; echo "b1 rc=$?"
timeout 600 python3 bench.py --tier e2e --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b8.json 2> $O/bench_e2e_b8.err; echo "b8 rc=$?"
timeout 600 python3 bench.py --tier e2e --workload txt2img --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_txt2img_b8.json 2> $O/bench_txt2img_b8.err; echo "txt2img rc=$?"
timeout 900 python3 bench.py --tier e2e --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b32.json 2> $O/bench_e2e_b32.err; echo "b32 rc=$?"
timeout 900 python3 bench.py --tier e2e --batch 32 --steps 1 --warmup 1 --image-stages vae+jpeg --no-cpu-baseline > $O/bench_sd21_jpeg_b32.json 2> $O/bench_sd21_jpeg_b32.err; echo "jpeg rc=$?"
timeout 900 python3 bench.py --tier e2e --batch 16 --steps 1 --warmup 1 --unet sd15 --height 768 --width 768 --no-cpu-baseline > $O/bench_sd15_768_b16.json 2> $O/bench_sd15_768_b16.err; echo "sd15 rc=$?"
timeout 300 python3 tools/unet_forward_bench.py 32 convs sd15 hw=96 > $O/unet_forward_sd15_768_b32_per_shape.txt 2>&1; head -3 $O/unet_forward_sd15_768_b32_per_shape.txt
timeout 300 python3 tools/unet_forward_bench.py 128 convs > $O/unet_forward_b128_per_shape.txt 2>&1; head -3 $O/unet_forward_b128_per_shape.txt
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -o e2e -- python3 $R/bench.py --tier e2e --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "rocprof rc=$?"
GSW_GRAPH=never timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sd15 -o sd15 -- python3 $R/bench.py --tier e2e --batch 16 --steps 1 --warmup 1 --unet sd15 --height 768 --width 768 --no-cpu-baseline > $O/bench_sd15_under_rocprof.json 2> $O/bench_sd15_under_rocprof.err; echo "rocprof sd15 rc=$?"
for rows in 1 16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rows$rows -o g -- python3 $R/tools/small_rows_profile.py $rows > $O/prof_rows$rows.log 2>&1; echo "graph rows $rows rc=$?"
  f=$(find $O/prof_rows$rows -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/forward_${rows}rows_graph_kernel_stats.csv
done
for rows in 128 64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${rows}_$c -o p -- python3 $R/tools/unet_forward_bench.py $rows > $O/pmc_${rows}_$c.log 2>&1; echo "pmc $rows $c rc=$?"
  done
  python3 $R/tools/pmc_traffic.py $O/pmc_${rows}_FETCH_SIZE $O/pmc_${rows}_WRITE_SIZE > $O/pmc_unet_forward_b${rows}_traffic.json
done
python3 $R/tools/pmc_family.py $O/pmc_unet_forward_b128_traffic.json $O/pmc_unet_forward_b64_traffic.json 64 > $O/e2e_dominant_kernel_pmc.json
cd $R
for d in prof_e2e prof_sd15; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O
python3 -c "
import json
d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['lossless'], d['roofline']['achieved'], d['roofline'].get('dense_tflops'), d['roofline'].get('conv3x3_tflops'), d['board'], d['tiers']['codec']['value'])
for f in ('bench_e2e_b1','bench_e2e_b8','bench_txt2img_b8','bench_e2e_b32','bench_sd21_jpeg_b32','bench_sd15_768_b16'):
    try:
        d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['lossless'], d['fallbacks_off_the_hand_written_path'])
    except Exception as e: print(f, 'FAILED', e)
"
head -8 $O/prof_e2e_kernel_stats.csv | cut -c1-160; head -8 $O/prof_sd15_kernel_stats.csv | cut -c1-160
