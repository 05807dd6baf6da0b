#!/bin/bash
# Round-5 job 15: the wide tile's epilogue split at its vmcnt(0) wait (cycle stamps), with and without the start stagger.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05o
mkdir -p $O
cd $R
for s in 0 4; do
echo "== GSW_MM_STAGGER=$s" >> $O/mm_trace_wide.txt
for a in "524288 320 2560 1 1 0" "131072 640 5120 1 1 0" "131072 640 1280 0 1 0" "524288 1280 320 0 1 1" "32768 5120 1280 0 1 1" "524288 320 640 0 1 0" "524288 320 320 0 1 1"; do
  GSW_MM_STAGGER=$s timeout 120 tools/ubench/bin/mm_trace_wide $a >> $O/mm_trace_wide.txt 2>&1
done
done
cat $O/mm_trace_wide.txt
