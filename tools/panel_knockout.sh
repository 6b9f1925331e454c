#!/bin/bash
# Timing-only knock-out builds of the row-panel GEMM (panel_kernel, K <= 256): each drops one part of the loop (results are wrong) and is
# timed on the train step's shapes.  Build here: bash tools/panel_knockout.sh build; on the GPU box: bash tools/panel_knockout.sh run
cd "$(dirname "$0")/.."
V="${VARIANTS:-BASE PKO_PLAIN_EPILOGUE PKO_NO_EPILOGUE PKO_ONE_MFMA PKO_NO_LDS_READS}"
if [ "$1" = build ]; then
  mkdir -p unast_amd/csrc/build_exp
  objs=$(ls unast_amd/csrc/build/*.o | grep -v -e /panel.o -e panel_stamps.o)
  for v in $V; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$v -c unast_amd/csrc/panel.hip -o unast_amd/csrc/build_exp/panel_$v.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs unast_amd/csrc/build_exp/panel_$v.o -ldl -o unast_amd/csrc/build_exp/libunast_$v.so || exit 1
  done
else
  for v in $V; do
    echo -n "$v: "; UNAST_HIP_LIB=$PWD/unast_amd/csrc/build_exp/libunast_$v.so python3 tools/panel_knockout.py
  done
fi
