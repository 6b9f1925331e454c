#!/bin/bash
# Same-box A/B of stream layouts and priorities: the eager step / the replay with captured-stream labels / the replay laid out from the DAG
# (the default; UNAST_REPLAY_LABELS=1 selects the labels), each with and without a high-priority speech stream (UNAST_STREAM_PRIO).
# usage (GPU box, repo root): bash tools/ab_labels.sh
run() {  # name launch labels prio
  UNAST_REPLAY_LABELS=$3 UNAST_STREAM_PRIO=$4 timeout -k 10 300 python bench.py --launch $2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); g = d.get('graph_replay') or {}
print('%-28s %7.3f ms/step   streams %s edges %s host %s' % ('$1', d['ms_per_step'], g.get('streams'), g.get('cross_stream_edges'), g.get('replay_host_ms', d.get('host_enqueue_ms_per_step'))))"
}
for rep in 1 2; do
  run "eager" eager 1 ""
  run "eager speech:-1" eager 1 "speech:-1"
  run "eager speech,speech_w:-1" eager 1 "speech:-1,speech_w:-1"
  run "replay labels" graph 1 ""
  run "replay labels speech:-1" graph 1 "speech:-1"
  run "replay dag" graph 0 ""
done
