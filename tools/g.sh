#!/bin/bash
# gpurun helper: `bash tools/g.sh <tag> <command...>` runs the command on the GPU box with gpurun_out/r4 present and its
# output in gpurun_out/r4/<tag>.log (gpurun_out/ does not travel to the box, so it has to be made there).
mkdir -p gpurun_out/r4
tag=$1; shift
bash -o pipefail -c "$*" > gpurun_out/r4/$tag.log 2>&1
rc=$?
tail -c 6000 gpurun_out/r4/$tag.log
exit $rc
