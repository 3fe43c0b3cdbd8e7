cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lzprof
rm -rf $O; mkdir -p $O
rocprofv3 --pmc VALUBusy SALUBusy --output-format csv -d $O/valu -- python3 $R/tools/lzbench.py 65536 4000 > $O/valu.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/sq -- python3 $R/tools/lzbench.py 65536 4000 > $O/sq.log 2>&1
