#!/bin/bash
# Round-4 measurement passes on the MI355X box (repo root).  usage: bash tools/run_r04_measure.sh <pass> [tag]
#   pass A: tests, the default bench line, configs 2 and 5, kernel statistics of the bench command (four streams / single stream)
#   pass B: HBM traffic and attention PMC passes, chip-idle trace of the replayed step, torch-native launch attribution, panel stamps and
#           micro-benchmarks.  Everything lands under gpurun_out/<tag>/; copy what is to be judged into profiles/.
pass=${1:-A}; tag=${2:-r4_final}; R=$PWD; O=$R/gpurun_out/$tag
mkdir -p $O
set -x
if [ "$pass" = "A" ]; then
  timeout -k 10 400 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log; tail -3 $O/tests.log
  timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  timeout -k 10 200 python bench.py --workload c2 --no-cpu-baseline > $O/c2.json 2> $O/c2.err
  timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --steps 10 --warmup 3 > $O/c5.json 2> $O/c5.err
  grep -o "\"ms_per_step\": [0-9.]*, \"higher" $O/*.json
  cd /tmp; export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_a -- python3 $R/bench.py --launch eager --no-cpu-baseline --steps 10 --warmup 3 > $O/prof_a.log 2>&1
  UNAST_SIDE_STREAMS=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -- python3 $R/bench.py --launch eager --no-cpu-baseline --steps 6 --warmup 2 > $O/prof_b.log 2>&1
  cd $R
else
  cd /tmp; export TMPDIR=/tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_replay -- python3 $R/bench.py --launch graph --no-cpu-baseline --steps 10 --warmup 3 > $O/trace_replay.log 2>&1
  cd $R
  python3 tools/trace_idle.py $O/trace_replay 8 > $O/idle.txt 2>&1; head -12 $O/idle.txt
  bash tools/pmc_hbm_traffic.sh ${tag}_pmc_hbm_traffic > $O/pmc_traffic.log 2>&1; tail -16 $O/pmc_traffic.log
  bash tools/pmc_attn.sh gpurun_out/$tag/pmc_attn > $O/pmc_attn.log 2>&1; grep -n "avg launch\|SQ_VALU_MFMA_BUSY\|GRBM_GUI\|SQ_INSTS_VALU\|SQ_INSTS_MFMA" $O/pmc_attn.log
  timeout -k 10 200 python tools/native_launches.py > $O/native_launches.txt 2>&1; tail -14 $O/native_launches.txt
  timeout -k 10 120 python tools/kpanel_stamps.py > $O/kpanel_stamps.txt 2>&1
  timeout -k 10 120 python tools/panel_stamps.py > $O/panel_stamps.txt 2>&1
  timeout -k 10 200 python tools/bench_kpanel.py > $O/bench_kpanel.txt 2>&1
  timeout -k 10 300 python tools/bench_panel.py > $O/bench_panel.txt 2>&1
  timeout -k 10 200 python tools/bench_attn.py > $O/bench_attn.txt 2>&1
  timeout -k 10 400 python tools/soak_graph.py > $O/soak_graph.txt 2>&1; tail -1 $O/soak_graph.txt
fi
