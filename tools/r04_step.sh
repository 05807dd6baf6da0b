# usage: bash tools/r04_step.sh <tag> [pytest args...]: GPU tests given, then the one- and two-row graph timelines + the small-batch probe
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ $# -gt 0 ]; then timeout 1500 python3 -m pytest "$@" -x -q -m gpu 2>&1 | tail -25 > $O/pytest.txt; tail -12 $O/pytest.txt; fi
cd /tmp && export TMPDIR=/tmp
for rows in 1 2; do
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tl_$rows -o t -- python3 $R/tools/small_rows_profile.py $rows > $O/tl_$rows.log 2>&1; echo "rocprof rows=$rows rc=$?"
python3 $R/tools/graph_timeline.py $O/tl_$rows 2 --full > $O/timeline_${rows}row.txt
done
cd $R
timeout 600 python3 tools/small_batch_probe.py 1 2 8 16 > $O/probe.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
grep -A60 "^replay:" $O/timeline_1row.txt | head -45
grep "^replay:" $O/timeline_2row.txt
cat $O/probe.txt | grep rows=
