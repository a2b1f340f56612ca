"""Print the dispatches of the last BFS in a rocprofv3 kernel-trace CSV: python tools/kt_print.py <csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
anchor = sys.argv[2] if len(sys.argv) > 2 else 'ResetKernel'
hits = [i for i, nm in enumerate(names) if anchor in nm]
idx = max(hits) if hits else max(0, len(rows) - 80)
t0 = int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    nm = r['Kernel_Name']
    short = nm.split('(')[0]
    short = short[:50] + ".." + short[-40:] if len(short) > 92 else short
    print("%8.1f us  dur %7.1f us  grid %8s wg %5s  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3,
          (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Grid_Size_X'], r['Workgroup_Size_X'], short))
