R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rows in 1 2; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$rows -o t -- python3 $R/tools/small_rows_profile.py $rows > $O/tl_$rows.log 2>&1; echo "rocprof rows=$rows rc=$?"
python3 $R/tools/graph_timeline.py $O/tl_$rows 2 --full > $O/timeline_${rows}row.txt
done
cd $R
timeout 600 python3 tools/small_batch_probe.py 1 2 8 16 shapes > $O/probe.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -45 $O/timeline_1row.txt
head -5 $O/probe.txt
