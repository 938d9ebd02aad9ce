#!/bin/bash
# SQ counters of the S2 activation kernels (tools/lab/s2_probe.py under rocprofv3 --pmc): where the wave cycles go.
# Two passes of four counters each (all eight do not fit the SQ counter slots of one pass on every build; tools/prof.sh
# splits FETCH_SIZE / WRITE_SIZE for the same reason).  A failed pass ends the script with the tail of its log.
#   bash tools/lab/s2_pmc.sh   ->  gpurun_out/s2_pmc/summary.txt
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/s2_pmc
raw=/tmp/s2_pmc_raw
mkdir -p $out $raw
pass() {   # pass <tag> <counters...>
  local tag=$1; shift
  if ! timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $raw/$tag -- python3 tools/lab/s2_probe.py > $out/run_$tag.log 2>&1; then
    echo "pmc pass $tag failed:"; tail -30 $out/run_$tag.log; exit 1
  fi
  if ! ls $raw/$tag/*/*counter_collection.csv > /dev/null 2>&1 && ! find $raw/$tag -name "*counter_collection.csv" | grep -q .; then
    echo "pmc pass $tag wrote no counter CSV:"; tail -30 $out/run_$tag.log; exit 1
  fi
}
pass a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass b SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
python3 - <<'PY'
import csv, glob, collections, sys
files = glob.glob("/tmp/s2_pmc_raw/**/*counter_collection.csv", recursive=True)
if not files:
    sys.exit("no counter CSV found under /tmp/s2_pmc_raw")
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "s2act" not in k:
            continue
        name = k.split("(anonymous namespace)::")[-1].split("(")[0] + " grid " + r.get("Grid_Size", r.get("Grid_Size_X", ""))
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[name] += 1
with open("gpurun_out/s2_pmc/summary.txt", "w") as o:
    for name, c in sorted(per.items()):
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        gui = c["GRBM_GUI_ACTIVE"] or 1.0
        line = (f"{name}: dispatches {n[name]}  VALU-issue share of wave cycles {c['SQ_ACTIVE_INST_VALU'] / wc:.3f}  any-instruction {c['SQ_ACTIVE_INST_ANY'] / wc:.3f}  "
                f"waiting (s_waitcnt/barrier) {c['SQ_WAIT_ANY'] / wc:.3f}  issue-stalled {c['SQ_WAIT_INST_ANY'] / wc:.3f}  "
                f"VALU instructions per wave-cycle-quad {c['SQ_INSTS_VALU'] / wc:.3f}  "
                f"VALU busy of SIMD time {4.0 * c['SQ_ACTIVE_INST_VALU'] / ((gui / 8.0) * 1024.0):.3f}")
        print(line); o.write(line + "\n")
PY
