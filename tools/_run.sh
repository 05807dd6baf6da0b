cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06p
timeout 900 python3 bench.py > gpurun_out/r06p/bench_default_box2.json 2> gpurun_out/r06p/bench_default_box2.err
python3 -c "
import json; d=json.load(open('gpurun_out/r06p/bench_default_box2.json')); print('box2 default', d['value'], d['board']['sclk_mhz_mean'], d['board']['power_w_mean'])"
bash tools/r06_profile.sh "prof small pmc codec sq xattn" 2>&1 | tail -40
