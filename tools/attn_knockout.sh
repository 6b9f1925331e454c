#!/bin/bash
# Timing-only knock-out builds of the forward attention kernel (attn_fwd32_kernel): each drops one part of the loop body (results are
# wrong) and is timed at 32 x 4 heads x 800 x 800, dropout 0.1, pre-split operands.  Build here: bash tools/attn_knockout.sh build
# (libraries under unast_amd/csrc/build_exp/); on the GPU box: bash tools/attn_knockout.sh run
cd "$(dirname "$0")/.."
V="${VARIANTS:-BASE KO_SOFTMAX KO_SMFMA KO_PVMFMA KO_BARRIER KO_GLOAD KO_LDS_STORE}"
if [ "$1" = build ]; then
  mkdir -p unast_amd/csrc/build_exp
  objs=$(ls unast_amd/csrc/build/*.o | grep -v -e attention.o -e panel_stamps.o)
  for v in $V; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -D$v -c unast_amd/csrc/attention.hip -o unast_amd/csrc/build_exp/attention_$v.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs unast_amd/csrc/build_exp/attention_$v.o -ldl -o unast_amd/csrc/build_exp/libunast_$v.so || exit 1
  done
else
  for v in $V; do
    echo -n "$v: "; UNAST_HIP_LIB=$PWD/unast_amd/csrc/build_exp/libunast_$v.so python3 tools/attn_knockout.py
  done
fi
