#!/bin/bash
# Timing-only knock-out builds of the LSTM forward recurrence (lstm_fwd_kernel): each drops one part of the step (results are wrong) and is
# timed by tools/bench_lstm.py at 64 sequences x 800 steps x 2 directions.  Build here: bash tools/lstm_knockout.sh build (libraries under
# unast_amd/csrc/build_exp/); on the GPU box: bash tools/lstm_knockout.sh run
cd "$(dirname "$0")/.."
V="${VARIANTS:-BASE LKO_DOT LKO_LDS LKO_ACT LKO_STORES LKO_BARRIER}"
if [ "$1" = build ]; then
  mkdir -p unast_amd/csrc/build_exp
  objs=$(ls unast_amd/csrc/build/*.o | grep -v -e lstm.o -e panel_stamps.o)
  for v in $V; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$v -c unast_amd/csrc/lstm.hip -o unast_amd/csrc/build_exp/lstm_$v.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs unast_amd/csrc/build_exp/lstm_$v.o -ldl -o unast_amd/csrc/build_exp/libunast_$v.so || exit 1
  done
else
  for v in $V; do
    echo -n "$v: "; UNAST_HIP_LIB=$PWD/unast_amd/csrc/build_exp/libunast_$v.so python3 tools/bench_lstm.py 2>&1 | grep lstm
  done
fi
