"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) for the scan kernel into
profiles/<tag>_pmc_traffic_scan_b<B>_10M.json.
    python tools/pmc_summary.py <fetch_dir> <write_dir> <B> <kernel_substr> [tag=r02] [bytes_per_element=4]"""
import csv, glob, json, sys
fetch_dir, write_dir, B, kern = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
tag = sys.argv[5] if len(sys.argv) > 5 else "r02"
esz = int(sys.argv[6]) if len(sys.argv) > 6 else 4
out = {}
for d, c in ((fetch_dir, 'FETCH_SIZE'), (write_dir, 'WRITE_SIZE')):
    f = max(glob.glob(f'{d}/**/*counter_collection.csv', recursive=True), key=__import__('os').path.getmtime)   # newest pass
    vals = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if kern in r['Kernel_Name'] and r['Counter_Name'] == c]
    out[c] = {'dispatches': len(vals), 'mean_KB': sum(vals) / len(vals), 'min_KB': min(vals), 'max_KB': max(vals)}
fetch = out['FETCH_SIZE']['mean_KB'] * 1024 * 2
write = out['WRITE_SIZE']['mean_KB'] * 1024
alg = 10_000_000 * 384 * esz
res = {'kernel': kern, 'workload': f'10,000,000 x 384, {esz} bytes per streamed element, {B} queries per launch',
       'command': f'rocprofv3 --pmc FETCH_SIZE --kernel-trace ... / rocprofv3 --pmc WRITE_SIZE --kernel-trace ... -- python3 bench.py --steps 5 --warmup 2 --batch {B} --no-cpu-baseline (two separate passes)',
       'raw': out, 'correction': 'FETCH_SIZE x2 on gfx950 for 16-B/lane coalesced streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as read',
       'hbm_read_bytes_per_launch': fetch, 'hbm_write_bytes_per_launch': write, 'hbm_bytes_per_launch': fetch + write,
       'algorithmic_bytes_per_launch': alg, 'ratio': (fetch + write) / alg}
json.dump(res, open(f'profiles/{tag}_pmc_traffic_scan_b{B}_10M.json', 'w'), indent=1)
print(B, kern, 'read', fetch / 1e9, 'write', write / 1e9, 'ratio', res['ratio'])
