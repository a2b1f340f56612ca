"""Aggregate rocprofv3 --pmc CSV output per kernel: python tools/pmc_summary.py <counter_collection.csv> <COUNTER>
Prints JSON {kernel: {"sum": counter sum, "dispatches": n}} (kernel names shortened to the function name)."""
import csv, json, re, sys

def short(name):
    name = name.split('(')[0]
    m = re.findall(r'([A-Za-z_][A-Za-z0-9_]*)\s*(?:<|$)', name.replace('void ', ''))
    parts = re.split(r'::', re.sub(r'<.*', '', name.replace('void ', '')))
    return parts[-1].strip() if parts else name

rows = csv.DictReader(open(sys.argv[1]))
want = sys.argv[2]
out = {}
for r in rows:
    if r.get('Counter_Name') != want:
        continue
    k = short(r['Kernel_Name'])
    d = out.setdefault(k, {"sum": 0.0, "dispatches": 0})
    d["sum"] += float(r['Counter_Value'])
    d["dispatches"] += 1
print(json.dumps(out, indent=1, sort_keys=True))
