cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06q
timeout 900 python3 bench.py > gpurun_out/r06q/bench_default.json 2> gpurun_out/r06q/bench_default.err
V=$(python3 -c "
import json; d=json.load(open('gpurun_out/r06q/bench_default.json')); print('box default', d['value'], d['board']['sclk_mhz_mean'], d['board']['power_w_mean'], d['board']['pci']); import sys; sys.exit(0 if d['value'] >= 8.08 else 1)")
rc=$?
echo "$V rc=$rc"
if [ $rc -eq 0 ]; then
  O=gpurun_out/r06q
  timeout 600 python3 bench.py --tier e2e --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b1.json 2> $O/bench_e2e_b1.err
  timeout 600 python3 bench.py --tier e2e --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b8.json 2> $O/bench_e2e_b8.err
  timeout 600 python3 bench.py --tier e2e --workload txt2img --batch 8 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_txt2img_b8.json 2> $O/bench_txt2img_b8.err
  timeout 900 python3 bench.py --tier e2e --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b32.json 2> $O/bench_e2e_b32.err
  timeout 900 python3 bench.py --tier e2e --batch 32 --steps 1 --warmup 1 --image-stages vae+jpeg --no-cpu-baseline > $O/bench_sd21_jpeg_b32.json 2> $O/bench_sd21_jpeg_b32.err
  timeout 900 python3 bench.py --tier e2e --batch 16 --steps 1 --warmup 1 --unet sd15 --height 768 --width 768 --no-cpu-baseline > $O/bench_sd15_768_b16.json 2> $O/bench_sd15_768_b16.err
  timeout 300 python3 tools/unet_forward_bench.py 128 convs > $O/unet_forward_b128_per_shape.txt 2>&1
  python3 - <<PY
import json
for f in ('bench_default', 'bench_e2e_b1', 'bench_e2e_b8', 'bench_txt2img_b8', 'bench_e2e_b32', 'bench_sd21_jpeg_b32', 'bench_sd15_768_b16'):
    d = json.load(open('gpurun_out/r06q/' + f + '.json')); print(f, round(d['value'], 3), round(d['ms_per_step'], 1), d['board']['sclk_mhz_mean'])
PY
fi
