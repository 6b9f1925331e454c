#!/bin/bash
run() { "$@" timeout -k 10 200 python bench.py --launch $MODE --steps 20 --warmup 5 --no-cpu-baseline --iso-steps 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); g=d.get('graph_replay') or {}; print(d['ms_per_step'], {k: g.get(k) for k in ('mode','kernels','memcpys','memsets','cross_stream_edges')})"; }
for i in 1 2; do
MODE=graph
echo "graph substeps: $(run env UNAST_JOINT_GEN=0)"
echo "graph joint: $(run env UNAST_JOINT_GEN=1)"
MODE=eager
echo "eager substeps: $(run env UNAST_JOINT_GEN=0)"
echo "eager joint: $(run env UNAST_JOINT_GEN=1)"
done
