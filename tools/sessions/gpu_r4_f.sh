#!/bin/bash
# round 4, GPU session F: (1) 1536-dimensional rows: 16-wave against 12-wave workgroups on one configs[4] shard; (2) index option
# sweep at 10M x 768 for bytes per query at recall 0.95; (3) HBM traffic and L2 hit rate of the build kernels at the driver shape
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out
export TMPDIR=/tmp
timeout 420 python tools/gpu_wide_row_probe.py 12500000 384 2>&1 | grep -v amdgpu | tee $O/r4f_wide_rows_1536.txt
timeout 900 python tools/gpu_option_sweep.py 10000000 32:64:256 24:48:256 24:48:384 32:64:384 16:32:128 2>&1 | grep -v amdgpu | tee $O/r4f_option_sweep_10m768.txt
P=$O/build_pmc_r04
mkdir -p $P
cd /tmp
for pass in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 400 rocprofv3 --kernel-trace --pmc $pass --kernel-include-regex "k_build_phase" -d $P/pmc_$tag -o pmc -- python $R/tools/gpu_build_probe.py 10000000 > $P/probe_$tag.txt 2> $P/pmc_$tag.err; echo "pmc $tag rc $?"
done
cd $R && python - "$P" <<'PY'
import json, os, sqlite3, sys
out = sys.argv[1]
res = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum} --kernel-include-regex k_build_phase -- "
                  "python tools/gpu_build_probe.py 10000000   (three passes; the driver shape: 10M x 768 cosine, M 32, ef_construction 256)"}
for line in open(os.path.join(out, "probe_FETCH_SIZE.txt")):
    if line.startswith("{"):
        res["probe"] = json.loads(line)
for tag, counters in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]), ("TCC_HIT_sum", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    try:
        d = sqlite3.connect(os.path.join(out, "pmc_%s" % tag, "pmc_results.db"))
        for c in counters:
            for name, n, total in d.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name = ? "
                                            "group by kernel_name", (c,)):
                key = "phase_a" if "phase_a" in name else "phase_b"
                res.setdefault(key, {})["kernel"] = name[:100]
                res[key]["launches"] = n
                res[key][c] = total
    except Exception as e:
        res["error_" + tag] = repr(e)
p = res.get("probe", {})
for key in ("phase_a", "phase_b"):
    k = res.get(key)
    if not k:
        continue
    k["hbm_bytes"] = k.get("FETCH_SIZE", 0) * 1024 * 2 + k.get("WRITE_SIZE", 0) * 1024
    if k.get("TCC_HIT_sum") is not None and k.get("TCC_MISS_sum"):
        k["l2_hit_rate"] = k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"])
if p and "phase_a" in res:
    res["phase_a"]["algorithmic_bytes"] = p["phase_a_algorithmic_bytes"]
    res["phase_a"]["fetched_over_algorithmic"] = res["phase_a"]["hbm_bytes"] / p["phase_a_algorithmic_bytes"]
if p and "phase_b" in res:
    alg_b = p["phase_b_distances"] * (4 * p["dim"] + 4)
    res["phase_b"]["algorithmic_bytes"] = alg_b
    res["phase_b"]["fetched_over_algorithmic"] = res["phase_b"]["hbm_bytes"] / alg_b
res["corrections"] = "bytes = counter * 1024; FETCH_SIZE doubled for 16-B/lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section)"
json.dump(res, open(os.path.join(os.path.dirname(out), "r4f_pmc_build_10m768.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $P/pmc_*
