#!/bin/bash
# A/B of the joint generator step's switches on the GPU box (same box, alternating): step time per launch mode.
run() { "$@" timeout -k 10 200 python bench.py --launch $MODE --steps 20 --warmup 5 --no-cpu-baseline --iso-steps 0 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); g=d.get('graph_replay') or {}; print(d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), {k: g.get(k) for k in ('mode','kernels','cross_stream_edges')})"; }
for i in 1 2; do
for MODE in eager graph; do
echo "$MODE substeps: $(run env UNAST_JOINT_GEN=0)"
echo "$MODE joint, decoder calls one by one: $(run env UNAST_JOINT_GEN=1 UNAST_JOINT_DECODERS=0)"
echo "$MODE joint, decoders paired: $(run env UNAST_JOINT_GEN=1 UNAST_JOINT_DECODERS=1)"
done
done
