set -x
B="timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch eager"
$B > gpurun_out/r2_b9_base.log 2> gpurun_out/r2_b9_base.err
UNAST_AUTOGRAD_ST=1 $B > gpurun_out/r2_b9_st.log 2> gpurun_out/r2_b9_st.err
UNAST_LN_FINALIZE_INLINE=1 $B > gpurun_out/r2_b9_lninl.log 2> gpurun_out/r2_b9_lninl.err
UNAST_WGRAD_STREAMS=0 $B > gpurun_out/r2_b9_nows.log 2> gpurun_out/r2_b9_nows.err
UNAST_WGRAD_GROUP_TARGET=768 $B > gpurun_out/r2_b9_t768.log 2> gpurun_out/r2_b9_t768.err
$B > gpurun_out/r2_b9_base2.log 2> gpurun_out/r2_b9_base2.err
UNAST_DDP_FORCE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b9_ddp.log 2> gpurun_out/r2_b9_ddp.err
grep -o "\"ms_per_step\": [0-9.]*, \"higher\|host_enqueue_ms_per_step\": [0-9.]*" gpurun_out/r2_b9_*.log
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_r02a -- python3 /root/repo/bench.py --launch eager --no-cpu-baseline --steps 10 --warmup 3 > /root/repo/gpurun_out/prof_r02a.log 2>&1
cd /root/repo
ls gpurun_out/prof_r02a/*/ | head
bash tools/pmc_hbm_traffic.sh r02_pmc_hbm_traffic > gpurun_out/r2_pmc_traffic.log 2>&1; tail -20 gpurun_out/r2_pmc_traffic.log
bash tools/pmc_attn.sh gpurun_out/pmc_attn_r02 > gpurun_out/r2_pmc_attn.log 2>&1; tail -60 gpurun_out/r2_pmc_attn.log
