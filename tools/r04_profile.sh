#!/bin/bash
# Round-4 evidence run on the GPU box (from the repo root).  Outputs under gpurun_out/r04p/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04p
mkdir -p $O
cd $R
timeout 1200 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python3 bench.py --tier e2e --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b1.json 2> $O/bench_e2e_b1.err; echo "b1 rc=$?"
timeout 600 python3 bench.py --tier e2e --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b8.json 2> $O/bench_e2e_b8.err; echo "b8 rc=$?"
timeout 900 python3 bench.py --tier e2e --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b32.json 2> $O/bench_e2e_b32.err; echo "b32 rc=$?"
timeout 900 python3 bench.py --tier e2e --batch 32 --steps 1 --warmup 1 --image-stages vae+jpeg --no-cpu-baseline > $O/bench_sd21_jpeg_b32.json 2> $O/bench_sd21_jpeg_b32.err; echo "jpeg rc=$?"
timeout 900 python3 bench.py --tier e2e --batch 16 --steps 1 --warmup 1 --unet sd15 --height 768 --width 768 --no-cpu-baseline > $O/bench_sd15_768_b16.json 2> $O/bench_sd15_768_b16.err; echo "sd15 rc=$?"
timeout 300 python3 bench.py --gpus 1 --preflight > $O/preflight_1gpu.json 2> $O/preflight_1gpu.err; echo "preflight rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_e2e -o e2e -- python3 $R/bench.py --tier e2e --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "rocprof rc=$?"
for rows in 1 2 16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rows$rows -o g -- python3 $R/tools/small_rows_profile.py $rows > $O/prof_rows$rows.log 2>&1; echo "rows $rows rc=$?"
done
for rows in 128 64; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${rows}_$c -o p -- python3 $R/tools/unet_forward_bench.py $rows > $O/pmc_${rows}_$c.log 2>&1; echo "pmc $rows $c rc=$?"
  done
  python3 $R/tools/pmc_traffic.py $O/pmc_${rows}_FETCH_SIZE $O/pmc_${rows}_WRITE_SIZE > $O/pmc_unet_forward_b${rows}_traffic.json
done
export GSW_MM_PANEL=8
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_128_panel8_FETCH -o p -- python3 $R/tools/unet_forward_bench.py 128 > $O/pmc_128_panel8.log 2>&1; echo "pmc panel8 rc=$?"
unset GSW_MM_PANEL
python3 $R/tools/pmc_traffic.py $O/pmc_128_panel8_FETCH $O/pmc_128_WRITE_SIZE > $O/pmc_unet_forward_b128_panel8_traffic.json
python3 $R/tools/pmc_family.py $O/pmc_unet_forward_b128_traffic.json $O/pmc_unet_forward_b64_traffic.json 64 > $O/e2e_dominant_kernel_pmc.json
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_codec_$c -o p -- python3 $R/bench.py --tier codec --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_codec_$c.log 2>&1; echo "pmc codec $c rc=$?"
done
python3 $R/tools/pmc_traffic.py $O/pmc_codec_FETCH_SIZE $O/pmc_codec_WRITE_SIZE > $O/pmc_codec_traffic.json
cd $R
for d in prof_e2e prof_rows1 prof_rows2 prof_rows16; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O; ls $O | head -50
python3 -c "
import json
d=json.load(open('$O/bench_default.json')); print('default', d['value'], d['ms_per_step'], d['lossless'], d['roofline']['achieved'], d['roofline'].get('dense_tflops'), d['roofline'].get('conv3x3_tflops'), d['cpu_baseline'], d['tiers']['codec']['value'], d['tiers']['codec']['cpu_baseline'])
for f in ('bench_e2e_b1','bench_e2e_b8','bench_e2e_b32','bench_sd21_jpeg_b32','bench_sd15_768_b16'):
    d=json.load(open('$O/'+f+'.json')); print(f, d['value'], d['ms_per_step'], d['lossless'], d['fallbacks_off_the_hand_written_path'])
print(open('$O/preflight_1gpu.json').read())
"
