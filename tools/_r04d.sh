cd $GRAFT_REPO_ROOT; O=gpurun_out/r04d; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest.txt; tail -5 $O/pytest.txt
bash tools/r04_step.sh r04d > $O/step.txt 2>&1; tail -40 $O/step.txt
timeout 900 python3 bench.py --tier e2e --batch 1 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b1.json 2> $O/bench_e2e_b1.err; tail -c 600 $O/bench_e2e_b1.json
timeout 900 python3 bench.py --tier e2e --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_e2e_b8.json 2> $O/bench_e2e_b8.err; tail -c 300 $O/bench_e2e_b8.json
