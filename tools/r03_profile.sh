# Round 3 profile set: rocprofv3 kernel stats + PMC passes for every kernel whose roofline the bench line quotes
# (DEGA encode / decode at cfg2, cfg3 = 1 Mi x 96, LZMH encode / decode at cfg4 full size).  Program directly after `--`.
#   usage (on the GPU box, from the repo root):  bash tools/r03_profile.sh [tag] [sections]
#   sections: any of  dega cfg3 lzmh  (default: all three)
set -x
TAG=${1:-r03}
SECTIONS=${2:-"dega cfg3 lzmh"}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}prof
mkdir -p $O
SQ="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
passes() {  # $1 = name, rest = program
    n=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${n}_stats -- "$@" > $O/${n}_stats.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${n}_fetch -- "$@" --no-round-trip > $O/${n}_fetch.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${n}_write -- "$@" --no-round-trip > $O/${n}_write.log 2>&1
    rocprofv3 --pmc VALUBusy SALUBusy --output-format csv -d $O/${n}_valu -- "$@" > $O/${n}_valu.log 2>&1
    rocprofv3 --pmc LDSBankConflict MeanOccupancyPerCU --output-format csv -d $O/${n}_lds -- "$@" > $O/${n}_lds.log 2>&1
    rocprofv3 --pmc $SQ --output-format csv -d $O/${n}_sq -- "$@" > $O/${n}_sq.log 2>&1
}
quick() {  # $1 = name, rest = program: kernel stats + the instruction-mix pass only
    n=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${n}_stats -- "$@" > $O/${n}_stats.log 2>&1
    rocprofv3 --pmc $SQ --output-format csv -d $O/${n}_sq -- "$@" > $O/${n}_sq.log 2>&1
    rocprofv3 --pmc VALUBusy SALUBusy --output-format csv -d $O/${n}_valu -- "$@" > $O/${n}_valu.log 2>&1
}
COMMON="--cpu-channels 0 --no-extras --end-to-end-channels 0"
for s in $SECTIONS; do
    case $s in
    dega) passes dega python3 $R/bench.py --steps 3 --warmup 1 $COMMON ;;
    cfg3) passes cfg3 python3 $R/bench.py --steps 3 --warmup 1 --channels 1048576 --samples 96 --step-size 300 $COMMON ;;
    dega_quick) quick dega python3 $R/bench.py --steps 3 --warmup 1 $COMMON ;;
    lzmh_quick) quick lzmh python3 $R/bench.py --workload lzmh --steps 2 --warmup 1 $COMMON ;;
    lzmh) passes lzmh python3 $R/bench.py --workload lzmh --steps 2 --warmup 1 $COMMON ;;
    esac
done
python3 $R/tools/r03_profile_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
