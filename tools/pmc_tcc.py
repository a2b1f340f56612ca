"""Fold rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum CSVs per kernel:
python tools/pmc_tcc.py <direction-optimizing counter_collection.csv> [<top-down-only counter_collection.csv>]
Prints JSON {leg: {kernel: {dispatches, hit, miss, hit_rate, requests_per_dispatch}}}.  TCC = the L2 of an XCD; a miss goes to
Infinity Cache / HBM."""
import csv, json, re, sys


def short(name):
    name = name.split('(')[0].replace('void ', '')
    return re.split(r'::', re.sub(r'<.*', '', name))[-1].strip()


def fold(path):
    per = {}
    seen = {}
    for r in csv.DictReader(open(path)):
        c = r.get('Counter_Name')
        if c not in ('TCC_HIT_sum', 'TCC_MISS_sum'):
            continue
        k = short(r['Kernel_Name'])
        d = per.setdefault(k, {"dispatches": 0, "hit": 0.0, "miss": 0.0})
        d["hit" if c == 'TCC_HIT_sum' else "miss"] += float(r['Counter_Value'])
        key = (k, r['Dispatch_Id'])
        if key not in seen:
            seen[key] = 1
            d["dispatches"] += 1
    for d in per.values():
        tot = d["hit"] + d["miss"]
        d["hit_rate"] = round(d["hit"] / tot, 4) if tot else None
        d["requests_per_dispatch"] = round(tot / max(d["dispatches"], 1))
    return per


out = {"direction_optimizing": fold(sys.argv[1])}
if len(sys.argv) > 2 and sys.argv[2]:
    out["top_down_only"] = fold(sys.argv[2])
print(json.dumps(out, indent=1, sort_keys=True))
