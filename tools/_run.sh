cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json | head -c 300; echo
timeout 900 python bench.py --tier e2e --no-cpu-baseline --unet sd15 --height 768 --width 768 --batch 16 > gpurun_out/bench_sd15.json 2> gpurun_out/bench_sd15.err
python - <<'PY'
import json
for n in ("default","sd15"):
    try:
        d=json.loads(open(f"gpurun_out/bench_{n}.json").read().strip().splitlines()[-1]); print(n, d["value"], d["ms_per_step"], d.get("board",{}).get("sclk_mhz_mean"))
    except Exception as e: print(n, "ERR", e)
PY
