#!/bin/bash
# HBM-side bytes per launch of the dominant kernels: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over
# `bench.py --steps 1 --warmup 0 --k 3`; writes gpurun_out/traffic.json (copy to profiles/traffic.json, which bench.py reads)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_traffic
rm -rf $OUT && mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 1 --warmup 0 --k 3 --no-cpu-baseline > $OUT/$c.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (bench.py --steps 1 --warmup 0 --k 3); counter unit KB; "
               "gfx950 correction: FETCH_SIZE doubled (wide 16-B/lane streams are tallied at half, MI355X_MICROARCH.md HBM section); per launch averages"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{out}/{c}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != c:
            continue
        for tag in ("k_rbf_fat_apply", "k_rbf_mfma_grad_h"):
            if tag in row["Kernel_Name"] and "false, false>" not in row["Kernel_Name"]:  # not the empty range-guard fallback launch
                acc[tag][0] += float(row["Counter_Value"]); acc[tag][1] += 1
    for tag, (v, n) in acc.items():
        res[f"{tag}_{c}_KB"] = v / n
        res[f"{tag}_launches_{c}"] = n
for tag in ("k_rbf_fat_apply", "k_rbf_mfma_grad_h"):
    if f"{tag}_FETCH_SIZE_KB" in res:
        res[f"{tag}_hbm_bytes_per_launch"] = (2 * res[f"{tag}_FETCH_SIZE_KB"] + res[f"{tag}_WRITE_SIZE_KB"]) * 1024
res["algorithmic_hbm_bytes_per_launch"] = 131072 * 8 * 4 + 2 * 64 * 131072 * 4 + 131072 * 4 * 9
json.dump(res, open(out + "/../traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
