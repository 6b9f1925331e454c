#!/bin/bash
# MFMA / VALU / LDS counters of the attention kernels at config-3 size (tools/prof_one.py attn_fwd | attn_bwd): one counter group
# per rocprofv3 run, --kernel-trace only.  usage (GPU box, repo root): bash tools/pmc_attn.sh <outdir>
out=${1:-gpurun_out/pmc_attn}; root=$PWD
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
groups=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
 "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM"
)
for which in attn_fwd attn_bwd; do
  i=0
  for g in "${groups[@]}"; do
    rocprofv3 --kernel-trace --pmc $g --output-format csv -d $root/$out/${which}_g$i -- python3 $root/tools/prof_one.py $which > $root/$out/${which}_g$i.log 2>&1 || echo "$which group $i failed" >> $root/$out/fail.log
    i=$((i+1))
  done
done
cd $root
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if "attn" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$out/*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if "attn" in k: dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open("$out/summary.txt", "w") as o:
    for k, d in agg.items():
        o.write("%s   (avg launch %.1f us under the counters)\n" % (k, sum(dur[k]) / max(len(dur[k]), 1) / 1e3))
        for c, v in sorted(d.items()):
            o.write("  %-32s n=%d mean=%.4g\n" % (c, len(v), sum(v) / len(v)))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CYCLES" in d:
            mf = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"]); bz = sum(d["SQ_BUSY_CYCLES"]) / len(d["SQ_BUSY_CYCLES"])
            o.write("  -> MFMA busy / (SQ busy cycles x 4 SIMDs): %.3f   (SQ_BUSY_CYCLES summed over SEs; see MI355X_MICROARCH.md for the normalisation)\n" % (mf / (bz * 4)))
print(open("$out/summary.txt").read())
PY
