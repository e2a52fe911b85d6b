"""Reads the rocprofv3 databases of the two PMC passes over scripts/ubench/fetch_calib and prints counter / bytes per
kernel (diagnostic; the result is kept as profiles/r03_counter_calibration.md)."""
import os, sqlite3, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/calib"
BYTES = float(1 << 30)
print("# FETCH_SIZE / WRITE_SIZE against known byte counts (scripts/ubench/fetch_calib.hip, 1 GiB per kernel)\n")
print("| kernel | access shape | counter | counter value (KB) | bytes moved | counter x 1024 / bytes |\n|---|---|---|---|---|---|")
shape = {"read16": "16-byte global_load per lane", "read8": "8-byte global_load per lane", "dma<16>": "LDS-DMA, 16-byte pieces",
         "dma<4>": "LDS-DMA, 4-byte pieces", "write16": "16-byte global_store per lane", "write8": "8-byte global_store per lane"}
for name, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    db = None
    for dp, _, fs in os.walk(os.path.join(root, sub)):
        for f in fs:
            if f.endswith(".db"):
                db = os.path.join(dp, f)
    if not db:
        print("| (no database for %s) |" % name)
        continue
    con = sqlite3.connect(db)
    for kn, cn, val in con.execute("select kernel_name, counter_name, value from counters_collection order by start"):
        key = next((k for k in shape if k.replace("<", "I").replace(">", "E") in kn or k in kn), None)
        if key is None:
            continue
        relevant = (name == "FETCH_SIZE") == (not key.startswith("write"))
        if relevant:
            print("| %s | %s | %s | %.1f | %d | %.3f |" % (key, shape[key], cn, val, int(BYTES), val * 1024.0 / BYTES))
