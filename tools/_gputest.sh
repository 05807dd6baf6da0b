#!/bin/bash
mkdir -p gpurun_out/gputest
python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/gputest/pytest_full.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3 >> gpurun_out/gputest/pytest_full.txt
cat gpurun_out/gputest/pytest_full.txt
bash tools/mm_ablate.sh run > gpurun_out/gputest/mm_power_ablation.txt 2>&1
cat gpurun_out/gputest/mm_power_ablation.txt
