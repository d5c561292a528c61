#!/usr/bin/env python3
"""profiles/traffic.json from a tools/profile.sh directory: HBM bytes per launch of the fused kernel
= 2 x FETCH_SIZE (gfx950 rocprofv3 reports half of the bytes of coalesced reads; calibrated with
tools/calib.hip, see profiles/r01_calibration.txt) + WRITE_SIZE, both in KiB."""
import csv, glob, json, os, sys
root, out = sys.argv[1], sys.argv[2]
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for f in glob.glob(os.path.join(root, "pmc*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "whisper400" in r["Kernel_Name"] and r["Counter_Name"] in vals:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"]) * 1024 * 2
write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"]) * 1024
json.dump({"clips": 1024, "clip_seconds": 10, "n_mels": 80, "hbm_bytes_per_launch": fetch + write,
           "fetch_bytes_corrected_x2": fetch, "write_bytes": write, "source": os.path.basename(root)},
          open(out, "w"), indent=1)
print(open(out).read())
