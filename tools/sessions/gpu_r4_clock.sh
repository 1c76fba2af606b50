#!/bin/bash
# round 4: does the shader clock differ between the visited sets in HBM and the compact sets in LDS?  (phase ticks fell 22 % per
# expansion while the launches got no shorter.)  GRBM_GUI_ACTIVE per dispatch / dispatch duration = busy cycles per ns.
ulimit -c 0
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=$PWD/gpurun_out
export TMPDIR=/tmp
cd /tmp
PROBE_PART_B_ONLY=1 timeout 150 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/clk -- python $GRAFT_REPO_ROOT/tools/gpu_compact_visited_probe.py 2000000 1536 > $O/r4_clock_probe.txt 2>&1
echo "rc $?"
python - <<'PY' | tee $O/r4_clock_summary.txt
import csv, glob
rows = []
for f in glob.glob("/tmp/clk/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
print("dispatches with counters:", len(rows))
ks = [r for r in rows if "k_search" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
ks.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in ks:
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if dur > 5_000_000:  # the 10 x 1024-query launches
        print("k_search lds %6s scratch %3s  %.2f ms  GRBM_GUI_ACTIVE %.0f  -> %.3f busy cycles per ns (summed over the counter's instances)" % (
            r["LDS_Block_Size"], r["Scratch_Size"], dur / 1e6, float(r["Counter_Value"]), float(r["Counter_Value"]) / dur))
PY
grep -v amdgpu.ids $O/r4_clock_probe.txt | grep "ef " | cut -c1-200
