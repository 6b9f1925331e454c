#!/bin/bash
# Round-2 measurement pass on the MI355X box (repo root): tests, the default bench line, configs 2 and 5, then the rocprofv3
# summaries that are copied into profiles/ (kernel stats of the bench command, HBM traffic PMC passes, attention PMC passes).
# usage: bash tools/run_r02_measure.sh [tag]     (tag names the output files under gpurun_out/, default r2_final)
tag=${1:-r2_final}
set -x
python -m pytest tests -m gpu -q > gpurun_out/${tag}_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_tests.log; tail -3 gpurun_out/${tag}_tests.log
timeout -k 10 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/${tag}_c2.json 2> gpurun_out/${tag}_c2.err
timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/${tag}_c5.json 2> gpurun_out/${tag}_c5.err
timeout -k 10 300 python bench.py --no-cpu-baseline --launch eager --iso-detail > gpurun_out/${tag}_eager.json 2> gpurun_out/${tag}_eager.err
timeout -k 10 300 python bench.py --no-cpu-baseline --launch graph > gpurun_out/${tag}_graph.json 2> gpurun_out/${tag}_graph.err
grep -o "\"ms_per_step\": [0-9.]*, \"higher" gpurun_out/${tag}_*.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_${tag}_a -- python3 /root/repo/bench.py --launch eager --no-cpu-baseline --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_${tag}_a.log 2>&1
UNAST_SIDE_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_${tag}_b -- python3 /root/repo/bench.py --launch eager --no-cpu-baseline --steps 6 --warmup 2 > /root/repo/gpurun_out/prof_${tag}_b.log 2>&1
cd /root/repo
bash tools/pmc_hbm_traffic.sh ${tag}_pmc_hbm_traffic > gpurun_out/${tag}_pmc_traffic.log 2>&1; tail -16 gpurun_out/${tag}_pmc_traffic.log
bash tools/pmc_attn.sh gpurun_out/pmc_attn_${tag} > gpurun_out/${tag}_pmc_attn.log 2>&1; grep -n "avg launch\|MFMA busy\|SQ_VALU_MFMA_BUSY\|GRBM_GUI\|SQ_INSTS_VALU\|SQ_INSTS_MFMA" gpurun_out/${tag}_pmc_attn.log
