#!/bin/bash
# Same-box A/B of the capture's dependency pruning (UNAST_CAPTURE_PRUNE=1, default: a backward segment keeps, of what its stream inherited
# through the origin stream's relay, only the producers of its own incoming gradients) against the unpruned capture, and the eager step.
# usage (GPU box, repo root): bash tools/ab_prune.sh
run() {
  UNAST_CAPTURE_PRUNE=$3 timeout -k 10 300 python bench.py --launch $2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); g = d.get('graph_replay') or {}
print('%-34s %7.3f ms/step   edges %s' % ('$1', d['ms_per_step'], g.get('cross_stream_edges')))"
}
for rep in 1 2; do
  run "replay, pruned capture" graph 1
  run "replay, unpruned capture" graph 0
  run "eager" eager 1
done
