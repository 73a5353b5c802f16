"""Parse rocprofv3 --pmc counter CSVs (FETCH_SIZE / WRITE_SIZE passes) into per-launch HBM bytes per
kernel, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section): counters are in KiB,
FETCH_SIZE reports exactly 1/2 of the bytes of wide (16 B/lane) coalesced reads -> doubled.
usage: pmc_traffic.py <dir with *counter_collection.csv> [<dir> ...] > profiles/r02_pmc_traffic.json"""
import csv, glob, json, os, sys
acc = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name") or row.get("Kernel Name")
            cn, cv = row.get("Counter_Name"), float(row.get("Counter_Value", 0))
            short = name.split("(")[0].replace("void ", "").replace("mpsk::", "").replace(" ", "")   # bench.py's kernel key
            a = acc.setdefault(short, {}).setdefault(cn, [0.0, 0])
            a[0] += cv; a[1] += 1
out = {}
for k, v in acc.items():
    fetch = v.get("FETCH_SIZE", [0, 1]); write = v.get("WRITE_SIZE", [0, 1])
    rd = 2.0 * fetch[0] / max(fetch[1], 1) * 1024.0
    wr = write[0] / max(write[1], 1) * 1024.0
    out[k] = rd + wr                                        # what bench.py puts into roofline.traffic
    out[k + "/detail"] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                          "launches_sampled": max(fetch[1], write[1])}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out["kernel_source_sha256_16"] = bench.kernel_source_hash()      # bench.py attaches the figures only to these sources
print(json.dumps(out, indent=1, sort_keys=True))
