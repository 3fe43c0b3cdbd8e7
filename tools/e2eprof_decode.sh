# kernel + memory-copy trace of the host-pointer DECODE pipeline (see e2eprof.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/e2eprof_dec
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/e2e_probe.py ${1:-65536} ${2:-10800} ${3:-pinned} decode > $O/stats.log 2>&1
cat $O/stats.log | grep -i "decode\|error" | tail -5
python3 $R/tools/e2e_trace_summary.py $O/stats/*/
