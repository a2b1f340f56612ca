"""Per dispatch (in dispatch order) the counters of a rocprofv3 --pmc CSV: python tools/pmc_rows.py <counter_collection.csv>"""
import csv, sys, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    key = int(r['Dispatch_Id'])
    d = rows.setdefault(key, {"name": r['Kernel_Name'].split('(')[0][-60:], "c": {}})
    d["c"][r['Counter_Name']] = d["c"].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k in sorted(rows):
    d = rows[k]
    print("%6d %-62s %s" % (k, d["name"], "  ".join("%s=%.4g" % kv for kv in sorted(d["c"].items()))))
