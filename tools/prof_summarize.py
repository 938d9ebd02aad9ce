"""Condense rocprofv3 output (kernel_stats / kernel_trace / counter_collection CSVs) into small per-kernel summaries."""
import collections
import csv
import glob
import os
import sys

raw, out = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else None


def short(name):
    """'void (anonymous namespace)::kern<2, 2, false, 4>((anonymous namespace)::Segs, ...' -> 'kern<2, 2, false, 4>'"""
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, out_ = 0, []
    for ch in n:
        if ch == "<":
            depth += 1
        if ch == "(" and depth == 0:
            break
        if ch == ">":
            depth -= 1
        out_.append(ch)
    return "".join(out_)[:90]


def first(pat):
    g = glob.glob(os.path.join(raw, pat), recursive=True)
    return g[0] if g else None


f = first("trace/**/*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as o:
        w = csv.DictWriter(o, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows[:80])
f = first("trace/**/*kernel_trace.csv")
if f:  # per (kernel, grid) durations: the stats file mixes launches of different sizes under one name
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "anonymous namespace" in r["Kernel_Name"] and "at::" not in r["Kernel_Name"]:
            agg[(short(r["Kernel_Name"]), r["Grid_Size_X"], r["Workgroup_Size_X"], r["VGPR_Count"], r["SGPR_Count"])].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(out, "singa_kernels_by_grid.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "grid_x", "wg_x", "vgpr", "sgpr", "calls", "avg_us", "min_us", "max_us", "total_ms"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow(list(k) + [len(v), round(sum(v) / len(v) / 1e3, 2), round(min(v) / 1e3, 2), round(max(v) / 1e3, 2),
                                  round(sum(v) / 1e6, 3)])
    # one replayed step: everything dispatched between two consecutive adam_kernel launches of the timed region (the
    # steps of the region are identical replays; the one with the fewest dispatches in between is a pure replay + the next
    # batch's prepare on the side stream)
    tr = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
    marks = [i for i, t in enumerate(tr) if "adam_kernel" in t[2]]
    step = None
    for a, b in zip(marks, marks[1:]):
        if step is None or b - a < step[1] - step[0]:
            step = (a, b)
    if step:
        per = collections.defaultdict(lambda: [0, 0])
        for t in tr[step[0] + 1: step[1] + 1]:
            per[short(t[2]) if "anonymous namespace" in t[2] and "at::" not in t[2] else t[2][:90]][0] += 1
            per[short(t[2]) if "anonymous namespace" in t[2] and "at::" not in t[2] else t[2][:90]][1] += t[1] - t[0]
        replay = {"launches": step[1] - step[0], "kernel_ms": round(sum(v[1] for v in per.values()) / 1e6, 3),
                  "wall_ms": round((tr[step[1]][0] - tr[step[0]][0]) / 1e6, 3)}
        with open(os.path.join(out, "replayed_step_kernels.csv"), "w", newline="") as o:
            w = csv.writer(o)
            w.writerow(["kernel", "launches", "total_us", "share"])
            for k, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
                w.writerow([k, v[0], round(v[1] / 1e3, 1), round(v[1] / max(1, sum(x[1] for x in per.values())), 4)])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = first(f"pmc_{c}/**/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if ("anonymous namespace" in name and "at::" not in name) or "calib_copy" in name:
            agg[(short(name), r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Counter_Name", c))].append(
                float(r["Counter_Value"]))
    with open(os.path.join(out, f"pmc_{c}.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "grid", "counter", "dispatches", "avg_value", "min", "max"])
        for k, v in sorted(agg.items()):
            w.writerow(list(k) + [len(v), sum(v) / len(v), min(v), max(v)])

# ---- MFMA counters (one pass: SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F32, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE)
mfma = {}
replay = globals().get("replay")
f = first("pmc_MFMA/**/*counter_collection.csv")
if f:
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        name = short(r.get("Kernel_Name", ""))
        if "gemm_f32_kernel" in name or "cgemm3m_f32_kernel" in name:      # one name serves launches from 10 us to 3 ms: keep the grids apart
            name += f" grid {r.get('Grid_Size', r.get('Grid_Size_X', ''))}"
        per[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            calls[name] += 1
    rows = []
    for name, c in per.items():
        busy, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0)
        if busy <= 0 or gui <= 0:
            continue
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; the chip has 1024 SIMDs: utilisation = MFMA-busy cycles summed over
        # SIMDs / (elapsed cycles x 1024)
        util = busy / ((gui / 8.0) * 1024.0)
        rows.append((name, calls[name], busy, c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0), gui, util))
    rows.sort(key=lambda r: -r[2])
    with open(os.path.join(out, "pmc_MFMA.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "dispatches", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "GRBM_GUI_ACTIVE", "mfma_util"])
        w.writerows([(r[0], r[1], int(r[2]), int(r[3]), int(r[4]), round(r[5], 4)) for r in rows[:60]])
    for r in rows[:16]:
        mfma[r[0][:70]] = {"dispatches": r[1], "mfma_util": round(r[5], 4), "mops_f32": int(r[3])}

import hashlib
import json
h = hashlib.sha1()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for src in ("singa_amd/csrc/singa_hip.hip", "singa_amd/csrc/so3_index.h"):
    with open(os.path.join(root, src), "rb") as fh:
        h.update(fh.read())
json.dump({"workload": workload, "lib_sha": h.hexdigest()[:12], "replayed_step": replay,
           "mfma": {"counters": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), eager pass", "kernels": mfma}
                   if mfma else None}, open(os.path.join(out, "meta.json"), "w"), indent=1)
