"""Per-kernel HBM-side traffic of the K5 bf16 forward from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/k5_bench.py --precision bf16 -> profiles/r03_k5_pmc_traffic_bf16_256x512.json.
    python tools/k5_pmc.py <fetch_dir> <write_dir>
Unit / gfx950 correction as MI355X_MICROARCH.md prescribes (both counters in KB; FETCH_SIZE x 2 for 16-byte-per-lane streaming reads)."""
import csv, glob, json, os, sys
fetch_dir, write_dir = sys.argv[1], sys.argv[2]
res = {}
for d, c in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
    f = max(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c or not r["Kernel_Name"].split("(")[0].split()[-1].startswith(("ce_", "void ce_")) and "ce_" not in r["Kernel_Name"][:40]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        e = res.setdefault(name, {"FETCH_SIZE": [], "WRITE_SIZE": []})
        e[c].append(float(r["Counter_Value"]))
out = {}
for name, e in res.items():
    big = lambda v: [x for x in v if x >= 0.5 * max(v)] if v else []          # the full-size launches (the last layer's tail is small)
    fz, wz = big(e["FETCH_SIZE"]), big(e["WRITE_SIZE"])
    out[name] = {"launches_counted": len(fz), "FETCH_SIZE_raw_MB_per_full_size_launch": round(sum(fz) / max(len(fz), 1) * 1024 / 1e6, 1),
                 "hbm_read_MB_per_full_size_launch": round(sum(fz) / max(len(fz), 1) * 1024 * 2 / 1e6, 1),
                 "hbm_write_MB_per_full_size_launch": round(sum(wz) / max(len(wz), 1) * 1024 / 1e6, 1)}
doc = {"workload": "K5 bf16 forward, 256 pairs x 512 tokens = 131072 tokens, 6 layers (tools/k5_bench.py --precision bf16)",
       "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace ... / rocprofv3 --pmc WRITE_SIZE --kernel-trace ... -- python3 tools/k5_bench.py --precision bf16 (two separate passes)",
       "correction": "FETCH_SIZE x 2 on gfx950 for 16-B/lane streaming reads (MI355X_MICROARCH.md); WRITE_SIZE as read; KB -> bytes x 1024.  The raw FETCH_SIZE is kept beside the corrected figure: for ce_attention (K / V / Q rows of 64 B per head, read once per workgroup) the RAW figure is the one that matches the algorithmic 302 MB",
       "algorithmic_MB_per_layer": {"ce_ffn_fused (attention output + LN + FFN + LN)": "read ctx 100.7 + residual 201.3 (+ 3.5 of weights per workgroup from L2), write residual 201.3 + bf16 copy 100.7",
                                    "ce_proj_ts (QKV)": "read 100.7, write 302.0", "ce_attention": "read 302.0, write 100.7"},
       "kernels": out}
json.dump(doc, open("profiles/r03_k5_pmc_traffic_bf16_256x512.json", "w"), indent=1)
print(json.dumps(out, indent=1))
