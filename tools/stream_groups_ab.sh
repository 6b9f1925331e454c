# Same-box A/B of which logical streams share a real stream (4 real streams, 4 hardware queues unless noted).
run() { echo "$1 | HWQ=$2 | $(GPU_MAX_HW_QUEUES=$2 UNAST_STREAM_GROUPS="$3" timeout -k 10 300 python tools/host_vs_gpu.py 2>&1 | grep back-to-back)"; }
run "P1 default: text+disc_w share a queue" 4 ""
run "P2 text+disc | speech | speech_w | disc_w" 4 "text:q0,disc:q0"
run "P3 disc+disc_w | text | speech | speech_w" 4 "disc:q2,q2_w:q2"
run "P4 text+speech_w | speech | disc | disc_w" 4 "text:q0,speech_w:q0"
run "P5 disc+speech_w | text | speech | disc_w" 4 "disc:q2,speech_w:q2"
run "P6 speech+disc_w | text | disc | speech_w" 4 "speech:q1,disc_w:q1"
run "R  robust: text+disc+companions | speech | speech_w" 4 "text:q0,disc:q0,speech:q1,q1_w:q3,q0_w:q0"
run "R  robust, 8 queues" 8 "text:q0,disc:q0,speech:q1,q1_w:q3,q0_w:q0"
