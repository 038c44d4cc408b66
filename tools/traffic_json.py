#!/usr/bin/env python3
"""per-kernel HBM traffic json (what bench.py reads for roofline.traffic) from the per-kernel PMC table of tools/pmc_summary.py:
    python tools/traffic_json.py gpurun_out/final_r04_pmc_per_kernel.csv profiles/r04/traffic_r4.json
fetch_bytes = FETCH_SIZE (KB, as rocprofv3 prints it) x 1024 x 2 - the gfx950 correction of MI355X_MICROARCH.md's HBM section -, write_bytes =
WRITE_SIZE x 1024; per-dispatch averages of separate --pmc passes."""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = {'_note': 'per-dispatch averages from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/gpu_final.sh -> tools/gpu_pmc.sh, '
                'bench.py --steps 2 --in-flight 1); fetch_bytes = FETCH_SIZE KB x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section), '
                'write_bytes = WRITE_SIZE KB x 1024'}
for r in rows:
    if not r.get('FETCH_SIZE') or not r.get('WRITE_SIZE'):
        continue
    out[r['kernel']] = {'dispatches': int(float(r['dispatches_per_pass'])), 'avg_us_under_pmc': float(r['avg_us']), 'fetch_kb_raw': float(r['FETCH_SIZE']),
                        'fetch_bytes': float(r['FETCH_SIZE']) * 1024 * 2, 'write_bytes': float(r['WRITE_SIZE']) * 1024}
json.dump(out, open(sys.argv[2], 'w'), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != '_note'}, indent=1)[:1500])
