"""profiles/r03_k2_*: achieved GB/s of the BM25 kernels from a rocprofv3 --kernel-trace run of tools/k2_profile.py (+ the
FETCH_SIZE / WRITE_SIZE passes).    python tools/k2_summary.py <calls.json> <trace_dir> [fetch_dir] [write_dir] > out.json"""
import csv, glob, json, sys
calls = json.load(open(sys.argv[1]))
def newest(d, pat):
    fs = glob.glob(f"{d}/**/{pat}", recursive=True)
    return max(fs, key=len) if fs else None
rows = sorted(csv.DictReader(open(newest(sys.argv[2], "*kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
sl = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "rr_bm25_slices" in r["Kernel_Name"]]
at = {m: [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if f"rr_bm25_at<{m}>" in r["Kernel_Name"]] for m in (0, 1)}
out = {"workload": calls, "rr_bm25_slices": []}
gs = calls["get_scores_calls"]
per_call = len(sl) // max(len(gs), 1)          # (launches per get_scores call)
for i, c in enumerate(gs):
    us = sum(sl[i * per_call:(i + 1) * per_call])
    out["rr_bm25_slices"].append({"sum_df": c["sum_df"], "algorithmic_MB": round(c["algorithmic_bytes"] / 1e6, 1), "kernel_us": round(us, 1),
                                  "achieved_GBps": round(c["algorithmic_bytes"] / us / 1e3, 1), "frac_of_8TBps": round(c["algorithmic_bytes"] / us / 1e3 / 8000, 4)})
for m, name in ((0, "forward"), (1, "postings")):
    if at[m]:
        v = sorted(at[m])
        out[f"rr_bm25_at<{m}> ({name})"] = {"launches": len(v), "median_us": round(v[len(v) // 2], 2), "min_us": round(v[0], 2),
                                           "bound": "latency (150 x ~4 threads per query, ~6 dependent probes each)"}
def pmc(d, counter, kern):
    f = newest(d, "*counter_collection.csv")
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kern in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return vals
if len(sys.argv) > 3:
    v = pmc(sys.argv[3], "FETCH_SIZE", "rr_bm25_slices")
    out["FETCH_SIZE_KB_per_launch_raw"] = [round(x, 1) for x in v[:len(gs) * per_call]]
    out["FETCH_SIZE_note"] = ("raw counter (KB); the x2 correction of MI355X_MICROARCH.md is calibrated for 16-B-per-lane streaming reads only -- this "
                              "kernel reads 4-B words (coalesced 256 B per wave instruction): uncalibrated, read as a lower bound")
if len(sys.argv) > 4:
    v = pmc(sys.argv[4], "WRITE_SIZE", "rr_bm25_slices")
    out["WRITE_SIZE_KB_per_launch_raw"] = [round(x, 1) for x in v[:len(gs) * per_call]]
print(json.dumps(out, indent=1))
