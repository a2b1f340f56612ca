"""Counter values per dispatch of the LAST BFS in a rocprofv3 --pmc CSV: python tools/pmc_last_bfs.py <counter_collection.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r['Dispatch_Id']), {"name": r['Kernel_Name'], "c": collections.OrderedDict()})
    d["c"][r['Counter_Name']] = d["c"].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
ids = sorted(disp)
resets = [i for i in ids if 'ResetKernel' in disp[i]["name"]]
last = max(resets) if resets else ids[0]
for i in ids:
    if i < last:
        continue
    nm = disp[i]["name"].split('(')[0].replace('void ', '')
    nm = nm.split('<')[0].split('::')[-1]
    print("%-24s %s" % (nm, "  ".join("%s=%.4g" % kv for kv in disp[i]["c"].items())))
print()
