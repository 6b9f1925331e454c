#!/bin/bash
# HBM traffic per kernel launch for the bench command: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), kernel trace only,
# aggregated into profiles/<name>.json with the gfx950 FETCH_SIZE correction (x2; MI355X_MICROARCH.md, HBM section).
# usage (GPU box, repo root): bash tools/pmc_hbm_traffic.sh <out-json-basename>
name=${1:-pmc_hbm_traffic}; root=$PWD; out=$root/gpurun_out/$name
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --launch eager > $out/fetch.log 2>&1 || echo "fetch pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --launch eager > $out/write.log 2>&1 || echo "write pass failed"
cd $root
python3 - <<PY
import csv, glob, json, collections, re
def collect(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg
fe, wr = collect("$out/fetch", "FETCH_SIZE"), collect("$out/write", "WRITE_SIZE")
ks = {}
for k in fe:
    n = len(fe[k]); f = sum(fe[k]) / n * 1024 / 1e6 * 2.0        # counter unit: KiB; x2 on gfx950
    w = sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1) * 1024 / 1e6
    ks[k] = {"launches": n, "fetch_MB_per_launch_corrected_x2": round(f, 2), "write_MB_per_launch": round(w, 2), "hbm_MB_per_launch": round(f + w, 2)}
ks = dict(sorted(ks.items(), key=lambda kv: -kv[1]["hbm_MB_per_launch"] * kv[1]["launches"]))
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 3 --warmup 1 (config 3); per-launch averages over all launches of the kernel symbol; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md section HBM); values in MB (1e6 B)", "kernels": ks}, open("$out/$name.json", "w"), indent=1)
tot = sum(v["hbm_MB_per_launch"] * v["launches"] for v in ks.values()) / 6 / 1e3      # 1 warm-up + 3 timed + 2 single-stream steps
print("total HBM traffic per step ~ %.1f GB" % tot)
for k, v in list(ks.items())[:14]: print("%-60s" % k[:60], v)
PY
