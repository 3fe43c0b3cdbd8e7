# kernel + memory-copy trace of the host-pointer pipeline (which copies run on the copy engines, which as blit kernels)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/e2eprof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/e2e_probe.py ${1:-65536} ${2:-10800} ${3:-pinned} > $O/stats.log 2>&1
ls $O/stats/*/
cat $O/stats/*/*kernel_stats.csv | head -8
cat $O/stats/*/*memory_copy_stats.csv 2>/dev/null | head -8
python3 $R/tools/e2e_trace_summary.py $O/stats/*/
