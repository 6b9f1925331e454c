run() { echo "$1 | $(GPU_MAX_HW_QUEUES=4 UNAST_STREAM_GROUPS="$2" UNAST_WGRAD_COMPANION_OF="$3" timeout -k 10 300 python tools/host_vs_gpu.py 2>&1 | grep back-to-back)"; }
run "base: text | speech | disc | speech_w" "" "speech"
run "V1 text+speech | disc | speech_w" "text:q1,speech:q1" "q1"
run "V2 text | speech | disc+speech_w" "disc:q3,speech_w:q3" "speech"
run "V3 text+disc | speech | speech_w" "text:q0,disc:q0" "speech"
run "base again" "" "speech"
