run() { echo "$1 | HWQ=$2 | $(GPU_MAX_HW_QUEUES=$2 UNAST_STREAM_GROUPS="$3" timeout -k 10 300 python tools/host_vs_gpu.py 2>&1 | grep back-to-back)"; }
run "E0 default 7 logical streams" 4 ""
run "E1 text+disc_w | speech | disc | speech_w" 8 "text:q0,speech:q1,disc:q2,q1_w:q3,q2_w:q0,q0_w:q0"
run "E2 text+disc(+w) | speech | speech_w" 8 "text:q0,disc:q0,speech:q1,q1_w:q3,q0_w:q0"
run "E3 text+disc+speech_w | speech" 8 "text:q0,disc:q0,speech:q1,q1_w:q0,q0_w:q0"
run "E4 text | speech | disc | all wgrad on one" 8 "text:q0,speech:q1,disc:q2,q0_w:q3,q1_w:q3,q2_w:q3"
run "E2 again at HWQ=4" 4 "text:q0,disc:q0,speech:q1,q1_w:q3,q0_w:q0"
run "E1 again at HWQ=4" 4 "text:q0,speech:q1,disc:q2,q1_w:q3,q2_w:q0,q0_w:q0"
