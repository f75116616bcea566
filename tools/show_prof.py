import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
for r in rows[:n]:
    nm = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print('%-46s calls %5s avg %9.2f us  %5s%%' % (nm[:46], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
