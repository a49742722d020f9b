"""Sum a rocprofv3 --pmc counter_collection CSV per kernel name: python tools/pmc_summarize.py <csv> <out.json>"""
import csv, json, re, sys
from collections import defaultdict
rows = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
seen = set()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], name)
        if key not in seen:
            seen.add(key); calls[name] += 1
json.dump({k: {"calls": calls[k], **v} for k, v in rows.items()}, open(sys.argv[2], "w"), indent=1)
