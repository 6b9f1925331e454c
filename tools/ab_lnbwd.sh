#!/bin/bash
# Same-box A/B: LayerNorm backward in the epilogue of the input-gradient GEMM (UNAST_PANEL_LNBWD=1, default) against GEMM + stand-alone
# LayerNorm backward (0); stream replay and eager.  usage (GPU box, repo root): bash tools/ab_lnbwd.sh
run() {
  UNAST_PANEL_LNBWD=$3 timeout -k 10 300 python bench.py --launch $2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); g = d.get('graph_replay') or {}
print('%-40s %7.3f ms/step   kernels %s' % ('$1', d['ms_per_step'], g.get('kernels')))"
}
for rep in 1 2; do
  run "replay, LN backward fused" graph 1
  run "replay, LN backward stand-alone" graph 0
  run "eager, LN backward fused" eager 1
  run "eager, LN backward stand-alone" eager 0
done
