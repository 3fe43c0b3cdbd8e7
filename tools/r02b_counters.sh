set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
B="python3 $R/bench.py --steps 2 --warmup 1 --cpu-channels 0 --no-extras --end-to-end-channels 0"
rocprofv3 --pmc VALUBusy SALUBusy --output-format csv -d $O/valu -- $B > $O/valu.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq -- $B > $O/sq.log 2>&1
find $O -name "*counter_collection.csv" | head
