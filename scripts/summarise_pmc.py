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

# machine-readable copy for bench.py's roofline.traffic: mean KiB per dispatch of the log-likelihood kernel (the merged
# launch ppcx_ls_kernel of a pipelined round, or ppcx_loglik_kernel of the three-launch round)
import json
res = {}
for name, sub in [("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")]:
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == name and ("ppcx_ls_kernel" in row.get("Kernel_Name", "") or "loglik" in row.get("Kernel_Name", "")):
                    tot += float(row["Counter_Value"]); n += 1
    res[name.lower() + "_kib_per_launch"] = tot / max(n, 1)
    res[name.lower() + "_dispatches"] = n
res["chains_per_launch"] = int(os.environ.get("PPCX_PROFILE_CHAINS", "0"))
res["note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean over the log-likelihood kernel's dispatches of a "
               "shortened fit; raw counter units (KiB). gfx950: FETCH_SIZE under-reads wide (16 B/lane) coalesced streams by 2x; "
               "this kernel reads 4 B/lane, a width the guide calls uncalibrated, so the figure is a lower bound within 2x.")
json.dump(res, open(os.path.join(out, "pmc.json"), "w"), indent=1)
