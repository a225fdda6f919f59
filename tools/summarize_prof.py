"""Condense the rocprofv3 CSV output of tools/profile.sh into one text/JSON summary."""
import sys as _sys
if {"-h", "--help"} & set(_sys.argv[1:]):          # every tool answers --help without touching the GPU
    print(__doc__)
    raise SystemExit(0)
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(root, pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def main(out):
    summary = {"kernels": {}, "counters": {}}
    # timing pass: kernel_stats.csv / kernel_trace.csv
    for f in find(os.path.join(out, "trace"), "*kernel_stats.csv"):
        for row in csv.DictReader(open(f)):
            summary["kernels"][row["Name"]] = {k: row[k] for k in row if k != "Name"}
    durations = defaultdict(list)
    spans = defaultdict(list)
    for f in find(os.path.join(out, "trace"), "*kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            t0, t1 = int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            durations[row["Kernel_Name"]].append(t1 - t0)
            spans[row["Kernel_Name"]].append((t0, t1))
    for k, v in durations.items():
        v.sort()
        # busy_union_ns_per_call: the time during which AT LEAST ONE launch of this kernel was running, divided by the launches --
        # with independent frames in flight on two streams (bench.py --dispatch two_streams) two launches overlap, every one of them
        # lasts about twice the per-frame time, and this is the figure that corresponds to the bench line's launch_us
        iv = sorted(spans[k])
        union, cur0, cur1 = 0, iv[0][0], iv[0][1]
        for a0, a1 in iv[1:]:
            if a0 <= cur1:
                cur1 = max(cur1, a1)
            else:
                union += cur1 - cur0
                cur0, cur1 = a0, a1
        union += cur1 - cur0
        summary["kernels"].setdefault(k, {})["trace"] = {
            "calls": len(v), "avg_ns": sum(v) / len(v), "median_ns": v[len(v) // 2], "min_ns": v[0], "max_ns": v[-1],
            "busy_union_ns_per_call": union / len(v), "overlap_factor": sum(v) / union}
    # counter passes: counter_collection.csv, one row per (dispatch, counter)
    for d in sorted(os.listdir(out)):
        p = os.path.join(out, d)
        if not os.path.isdir(p) or d == "trace":
            continue
        acc = defaultdict(lambda: defaultdict(list))
        for f in find(p, "*counter_collection.csv"):
            for row in csv.DictReader(open(f)):
                acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for kern, cs in acc.items():
            for c, vals in cs.items():
                # skip the warm-up dispatches: average the last half
                tail = vals[len(vals) // 2:]
                summary["counters"].setdefault(kern, {})[c] = {"per_launch": sum(tail) / len(tail), "n": len(vals)}
    # calibration: kernels that move exactly 2^30 bytes each way -> bytes per reported KB
    calib = {}
    for kern, cs in summary["counters"].items():
        if kern.startswith("calib_"):
            for c, v in cs.items():
                if c in ("FETCH_SIZE", "WRITE_SIZE") and v["per_launch"] > 0:
                    calib.setdefault(c, {})[kern.split("(")[0]] = (2.0 ** 30) / (v["per_launch"] * 1024.0)
    summary["calibration_true_bytes_per_reported_byte"] = calib
    # HBM traffic of the unwarp kernels: reported KB x 1024 x calibration factor of the matching
    # access shape (dwordx2 gather for reads when available, else the coalesced dword copy)
    fr = calib.get("FETCH_SIZE", {})
    fw = calib.get("WRITE_SIZE", {})
    wf = fw.get("calib_copy_dword", 1.0)
    traffic = {}
    for kern, cs in summary["counters"].items():
        if "remap_" in kern or "stack_" in kern:
            # remap_lds_kernel streams 16-byte-per-lane row segments (the dwordx4 copy shape); the
            # direct kernels gather 8-byte tap pairs at 4-byte lane stride (the dwordx2 gather shape)
            if "remap_lds_kernel" in kern or "remap_wg_kernel" in kern or "stack_wg_kernel" in kern or "stack_lds_kernel" in kern:
                rf = fr.get("calib_copy_dwordx4", 2.0)
            else:
                rf = fr.get("calib_gather_dwordx2", fr.get("calib_copy_dword", 2.0))
            if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
                rd = cs["FETCH_SIZE"]["per_launch"] * 1024.0 * rf
                wr = cs["WRITE_SIZE"]["per_launch"] * 1024.0 * wf
                traffic[kern] = {"read_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                                 "read_factor": rf, "write_factor": wf}
    summary["traffic"] = traffic
    json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print("CALIBRATION (true bytes per reported byte):", json.dumps(calib))
    for k, v in traffic.items():
        print("TRAFFIC", k[:80], json.dumps(v))
    for k, v in summary["kernels"].items():
        print("KERNEL", k[:100])
        for kk, vv in v.items():
            print("   ", kk, vv)
    for kern, cs in summary["counters"].items():
        print("COUNTERS", kern[:100])
        for c in sorted(cs):
            print("    %-44s %18.1f  (n=%d)" % (c, cs[c]["per_launch"], cs[c]["n"]))


if __name__ == "__main__":
    main(sys.argv[1])
