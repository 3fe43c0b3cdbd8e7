cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/e2eprof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/e2e_probe.py 65536 10800 pinned > $O/stats.log 2>&1
cat $O/stats/*/*kernel_stats.csv | head -12
