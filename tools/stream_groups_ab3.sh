run() { echo "$1 | $(env $2 timeout -k 10 300 python tools/host_vs_gpu.py 2>&1 | grep back-to-back)"; }
run "base" "UNAST_X=0"
run "text wgrads -> speech_w (min tokens 5000)" "UNAST_WGRAD_COMPANION_OF=speech,text UNAST_STREAM_GROUPS=text_w:speech_w UNAST_WGRAD_MIN_TOKENS=5000"
run "base" "UNAST_X=0"
run "text wgrads -> speech_w (min tokens 5000)" "UNAST_WGRAD_COMPANION_OF=speech,text UNAST_STREAM_GROUPS=text_w:speech_w UNAST_WGRAD_MIN_TOKENS=5000"
run "disc wgrads too -> speech_w" "UNAST_WGRAD_COMPANION_OF=speech,text,disc UNAST_STREAM_GROUPS=text_w:speech_w,disc_w:speech_w UNAST_WGRAD_MIN_TOKENS=5000"
