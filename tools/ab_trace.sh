#!/bin/bash
# kernel-trace timeline of one step (tools/trace_timeline.py): usage ab_trace.sh <graph|eager> [env assignments...]
R=$PWD; O=$R/gpurun_out/r4/abt; mkdir -p $O
mode=${1:-eager}; shift
cd /tmp; export TMPDIR=/tmp
env "$@" timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$mode -- python3 $R/bench.py --launch $mode --no-cpu-baseline --steps 8 --warmup 3 --iso-steps 0 > $O/tr_$mode.log 2>&1
python3 $R/tools/trace_timeline.py $O/tr_$mode 2
