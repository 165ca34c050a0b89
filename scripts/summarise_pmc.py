"""Average FETCH_SIZE / WRITE_SIZE per dispatch of each kernel from rocprofv3 --pmc CSVs (development aid)."""
import csv, glob, os, sys, collections
out = sys.argv[1]
for name, sub in [("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")]:
    files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != name:
                    continue
                k = row.get("Kernel_Name", "?")
                k = k.split("(")[0][:60]
                acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    print(f"== {name} (raw counter units as reported by rocprofv3; gfx950: FETCH_SIZE x2 for wide coalesced reads, guide MI355X_MICROARCH.md HBM section)")
    for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"{k:60s} dispatches {n:8d}  mean/dispatch {s/max(n,1):14.2f}")
