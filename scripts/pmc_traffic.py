"""HBM traffic per launch of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE):
    python scripts/pmc_traffic.py <workload> <rows> <kernel substring> <fetch csv> <write csv>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a coalesced streaming read
(MI355X_MICROARCH.md, HBM section), so it is doubled.  Updates profiles/traffic.json."""
import collections, csv, json, sys
from pathlib import Path

wl, rows, kern, fcsv, wcsv = sys.argv[1:6]


def avg(path, counter):
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kern in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return sum(v) / len(v), len(v)


f, nf = avg(fcsv, "FETCH_SIZE")
w, nw = avg(wcsv, "WRITE_SIZE")
out = Path("profiles/traffic.json")
d = json.loads(out.read_text()) if out.exists() else {}
d[f"{wl}:{rows}"] = {"kernel": kern, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "launches": [nf, nw],
                     "hbm_bytes": int(2 * f * 1024 + w * 1024)}
out.write_text(json.dumps(d, indent=1, sort_keys=True) + "\n")
print(d[f"{wl}:{rows}"])
