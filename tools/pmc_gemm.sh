#!/bin/bash
# PMC passes for one GEMM shape (tools/prof_one.py gemm_fwd); one counter group per run, --kernel-trace only.
# usage (on the GPU box, from the repo root): bash tools/pmc_gemm.sh <which> <outdir>
which=${1:-gemm_fwd}; out=${2:-gpurun_out/pmc_gemm}; root=$PWD
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
 "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d $root/$out/g$i -- python3 $root/tools/prof_one.py $which > $root/$out/g$i.log 2>&1 || echo "group $i failed" >> $root/$out/fail.log
  i=$((i+1))
done
cd $root
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$out/summary.txt", "w") as o:
    for k, d in agg.items():
        if "gemm" not in k and "attn" not in k and "panel" not in k: continue
        o.write(k + "\n")
        for c, v in sorted(d.items()):
            o.write("  %-36s n=%d mean=%.4g\n" % (c, len(v), sum(v) / len(v)))
print(open("$out/summary.txt").read())
PY
