#!/bin/bash
# Round-2 measurement pass on the MI355X box (repo root): tests, the default bench line, configs 2 and 5, then the rocprofv3
# summaries that are copied into profiles/ (kernel stats of the bench command, HBM traffic PMC passes, attention PMC passes).
set -x
python -m pytest tests -m gpu -q > gpurun_out/r2_final_tests.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_final_tests.log; tail -3 gpurun_out/r2_final_tests.log
timeout -k 10 600 python bench.py > gpurun_out/r2_final_bench.json 2> gpurun_out/r2_final_bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline > gpurun_out/r2_final_c2.json 2> gpurun_out/r2_final_c2.err
timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --steps 10 --warmup 3 > gpurun_out/r2_final_c5.json 2> gpurun_out/r2_final_c5.err
timeout -k 10 300 python bench.py --no-cpu-baseline --launch eager --iso-detail > gpurun_out/r2_final_eager.json 2> gpurun_out/r2_final_eager.err
timeout -k 10 300 python bench.py --no-cpu-baseline --launch graph > gpurun_out/r2_final_graph.json 2> gpurun_out/r2_final_graph.err
grep -o "\"ms_per_step\": [0-9.]*, \"higher" gpurun_out/r2_final_*.json
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r02b -- python3 /root/repo/bench.py --launch eager --no-cpu-baseline --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_r02b.log 2>&1
UNAST_SIDE_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r02c -- python3 /root/repo/bench.py --launch eager --no-cpu-baseline --steps 6 --warmup 2 > /root/repo/gpurun_out/prof_r02c.log 2>&1
cd /root/repo
bash tools/pmc_hbm_traffic.sh r02_pmc_hbm_traffic > gpurun_out/r2_pmc_traffic.log 2>&1; tail -16 gpurun_out/r2_pmc_traffic.log
bash tools/pmc_attn.sh gpurun_out/pmc_attn_r02 > gpurun_out/r2_pmc_attn.log 2>&1; grep -n "avg launch\|MFMA busy\|SQ_VALU_MFMA_BUSY\|GRBM_GUI\|SQ_INSTS_VALU\|SQ_INSTS_MFMA" gpurun_out/r2_pmc_attn.log
