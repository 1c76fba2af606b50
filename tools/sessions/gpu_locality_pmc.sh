#!/bin/bash
# round 3: TCC hit/miss and FETCH_SIZE of k_search before / after the (level, cluster) reordering and with neighbouring queries
# handed out together (separate --pmc passes, as MI355X_MICROARCH.md prescribes)
ulimit -c 0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r03_locality
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass --kernel-include-regex k_search -d $O/pmc_$tag -o pmc -- python $R/tools/gpu_locality_pmc_probe.py 10000000 > $O/probe_$tag.txt 2> $O/probe_$tag.err
  tail -n 5 $O/probe_$tag.txt
done
cd $R && python - <<'PY'
import sqlite3, json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/prof_r03_locality"
out = {"command": "rocprofv3 --kernel-trace --pmc {TCC_HIT_sum TCC_MISS_sum | FETCH_SIZE | WRITE_SIZE} --kernel-include-regex k_search -- "
                  "python tools/gpu_locality_pmc_probe.py 10000000",
       "workload": "10M x 768 cosine, M 32, ef_construction 256, ef_search 96, 16 batches of 1024 queries per launch, one launch at a time",
       "launch_groups": {"A insertion order": [1, 2, 3], "B (level, cluster) order": [5, 6, 7], "C reordered + queries sorted by mixture component": [10, 11, 12]},
       "corrections": "FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled for 16-B/lane coalesced reads on gfx950 (MI355X_MICROARCH.md, HBM section)"}
vals = {}
for tag, names in (("TCC_HIT_sum", ("TCC_HIT_sum", "TCC_MISS_sum")), ("FETCH_SIZE", ("FETCH_SIZE",)), ("WRITE_SIZE", ("WRITE_SIZE",))):
    try:
        d = sqlite3.connect(O + "/pmc_%s/pmc_results.db" % tag)
        cols = [r[1] for r in d.execute("pragma table_info(counters_collection)").fetchall()]
        open(O + "/schema.txt", "w").write(" ".join(cols) + "\n" + repr(d.execute("select * from counters_collection limit 3").fetchall()))
        for name in names:
            rows = d.execute("select dispatch_id, sum(value) from counters_collection where counter_name = ? and kernel_name like '%k_search%' group by dispatch_id order by dispatch_id", (name,)).fetchall()
            vals[name] = [r[1] for r in rows]
    except Exception as e:
        print(tag, "failed", e)
out["per_launch"] = vals
for group, idxs in out["launch_groups"].items():
    g = {}
    for name, v in vals.items():
        if len(v) > max(idxs):
            g[name] = sum(v[i] for i in idxs) / len(idxs)
    if "TCC_HIT_sum" in g and "TCC_MISS_sum" in g:
        g["l2_hit_rate"] = g["TCC_HIT_sum"] / (g["TCC_HIT_sum"] + g["TCC_MISS_sum"])
    if "FETCH_SIZE" in g:
        g["hbm_read_bytes_per_launch"] = g["FETCH_SIZE"] * 1024 * 2
    out[group] = g
    print(group, json.dumps(g))
for tag in ("TCC_HIT_sum",):
    out["probe_output"] = open(O + "/probe_%s.txt" % tag).read().strip().splitlines()[-5:]
json.dump(out, open(O + "/locality_pmc.json", "w"), indent=1)
PY
[ -s $O/locality_pmc.json ] && grep -q TCC_HIT_sum $O/locality_pmc.json && rm -rf $O/pmc_*
du -sh $O
