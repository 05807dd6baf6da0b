#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05k
mkdir -p $O
cd $R
timeout 300 python3 tools/copy_sites.py 1 > $O/copy_sites_1row.txt 2>&1; head -70 $O/copy_sites_1row.txt
timeout 300 python3 tools/copy_sites.py 16 > $O/copy_sites_16rows.txt 2>&1; head -40 $O/copy_sites_16rows.txt
timeout 300 python3 tools/small_batch_probe.py > $O/small_batch_probe.txt 2>&1; tail -20 $O/small_batch_probe.txt
