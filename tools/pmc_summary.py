"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch) -> JSON.
usage: pmc_summary.py <name=dir> ...   e.g. fetch=gpurun_out/p_fetch write=gpurun_out/p_write"""
import collections, csv, glob, json, os, sys

out = {}
for arg in sys.argv[1:]:
    tag, d = arg.split("=", 1)
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        n = max(len(disp[k]), 1)
        out.setdefault(k, {})["dispatches_" + tag] = n
        for c, val in v.items():
            out[k][c + "_per_dispatch"] = val / n
print(json.dumps(out, indent=1))
