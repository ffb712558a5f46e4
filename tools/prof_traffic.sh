#!/bin/bash
# Memory-side bytes per launch of the dominant kernels: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over
# `bench.py --steps 1 --warmup 0 --k $K` (K = 40: the config-4 batch of 2560 rows in the gradient GEMM; round 4's file was collected
# with --k 3 on an intermediate build); writes gpurun_out/traffic.json (copy to profiles/traffic.json, which bench.py reads and names
# in `roofline.traffic_source`).  The GPU box has no .git: pass the commit the tree was built from,
#     COMMIT=$(git rev-parse --short HEAD) gpurun -- "COMMIT=$COMMIT tools/prof_traffic.sh"
# tests/test_host_cpu.py fails when that commit is older than the last change to the kernels it describes.
set -e
K=${K:-40}
COMMIT=${COMMIT:-unknown}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_traffic
rm -rf $OUT && mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -- python3 $R/bench.py --steps 1 --warmup 0 --k $K --no-cpu-baseline --no-modes > $OUT/$c.log 2>&1
done
python3 - $OUT $K $COMMIT <<'PY'
import csv, glob, json, sys, collections
out, k, commit = sys.argv[1:4]
res = {"commit": commit, "command": f"bench.py --steps 1 --warmup 0 --k {k} --no-cpu-baseline --no-modes", "krylov_depth": int(k),
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; counter unit KB; per launch averages.  gfx950 correction: "
               "FETCH_SIZE doubled (wide 16-B/lane streams are tallied at half, MI355X_MICROARCH.md HBM section).  Both counters sit on the "
               "L2's memory side (fabric requests: L2 misses), Infinity-Cache hits INCLUDED -- `hbm_bytes_per_launch` = 2 FETCH + WRITE is an "
               "upper bound of what reaches HBM, exact only for streams that miss the 256 MiB Infinity Cache"}
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
res["k_rbf_mfma_grad_h_algorithmic_bytes_per_launch"] = 2 * 2 * (64 * int(k)) * 131072 * 2   # hi + lo f16 images of L and R
res["k_rbf_mfma_grad_h_note"] = ("batch = 64 probes x k steps (2560 rows at k = 40: the config-4 sweep); memory-side bytes (2 x FETCH + WRITE) are ~90 x the operand "
                                 "images: every 256 x 256 tile re-reads its operand panels through L2 (hit rate 81 %, profiles/r04g_*), the misses are served by the "
                                 "Infinity Cache and, the 2.7 GB of images exceeding it, partly by HBM; WRITE_SIZE ~ 3 MB: the partial sums only, no scratch")
json.dump(res, open(out + "/../traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
