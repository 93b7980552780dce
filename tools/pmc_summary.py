"""Folds rocprofv3 --pmc counter_collection CSVs (one pass with FETCH_SIZE, one with WRITE_SIZE per workload) into
profiles/<round>_pmc_traffic.json.  HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in
KiB, and on gfx950 FETCH_SIZE counts each 128-byte request as 64 bytes.

usage: pmc_summary.py <out.json> <workload>=<fetch.csv>,<write.csv> ..."""
import csv, json, re, sys
from collections import defaultdict

BENCH_NAMES = {"k_agg_fwd": "aggregate_fwd", "k_agg_bwd_src": "aggregate_bwd_src"}


def read(path, counter):
    per = defaultdict(lambda: defaultdict(float))   # kernel -> dispatch id -> value (summed over instances)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            per[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in per.items()}


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1`; "
                 "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE counts 128-B requests as 64 B on gfx950)",
       "dtype": "f32", "kernels": {}}
for spec in sys.argv[2:]:
    wl, paths = spec.split("=")
    f, w = paths.split(",")
    fetch, write = read(f, "FETCH_SIZE"), read(w, "WRITE_SIZE")
    ks = {}
    for k in fetch:
        if "mpnn::" not in k:
            continue
        fa, n = fetch[k]
        wa = write.get(k, (0.0, 0))[0]
        rec = {"FETCH_SIZE_KB_avg": fa, "launches": n, "WRITE_SIZE_KB_avg": wa, "hbm_bytes_per_launch": (2 * fa + wa) * 1024}
        for pat, bname in BENCH_NAMES.items():
            if re.search(r"mpnn::" + pat + r"(_masked)?<", k):
                rec["bench_kernel"] = bname
        ks[short(k)] = rec
    out["kernels"][wl] = ks
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1])
