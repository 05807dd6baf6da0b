#!/bin/bash
# Round-5 job 1 on the GPU box (from the repo root): evidence for configs[4] (SD 1.5, 768 x 768) that no earlier round collected, and the codec tier's
# rocprofv3 kernel-duration summary.  Outputs under gpurun_out/r05a/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
timeout 600 python3 tools/attn_shapes_bench.py 32 all zeros > $O/attn_shapes_b32.txt 2>&1; echo "attn shapes rc=$?"
timeout 600 python3 tools/unet_forward_bench.py 32 convs sd15 hw=96 > $O/unet_forward_sd15_768_b32_per_shape.txt 2>&1; echo "sd15 per-shape rc=$?"
timeout 600 python3 tools/unet_forward_bench.py 128 convs > $O/unet_forward_b128_per_shape.txt 2>&1; echo "sd21 per-shape rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sd15 -o e2e -- python3 $R/bench.py --tier e2e --batch 16 --steps 1 --warmup 1 --unet sd15 --height 768 --width 768 --no-cpu-baseline > $O/bench_sd15_768_b16_under_rocprof.json 2> $O/bench_sd15_under_rocprof.err; echo "rocprof sd15 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_codec -o codec -- python3 $R/bench.py --tier codec --no-cpu-baseline > $O/bench_codec_under_rocprof.json 2> $O/bench_codec_under_rocprof.err; echo "rocprof codec rc=$?"
cd $R
for d in prof_sd15 prof_codec; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O; cat $O/attn_shapes_b32.txt; head -30 $O/unet_forward_sd15_768_b32_per_shape.txt; head -12 $O/prof_sd15_kernel_stats.csv; head -8 $O/prof_codec_kernel_stats.csv
