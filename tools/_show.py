import csv, json, glob, sys
d = sys.argv[1]
for f in sorted(glob.glob(f'gpurun_out/{d}/bench*.json')):
    try:
        j = json.load(open(f)); print(f, round(j['value']), round(j['ms_per_step'], 2))
    except Exception as e: print(f, 'ERR', e)
try:
    rows = list(csv.DictReader(open(f'gpurun_out/{d}/kernel_stats.csv')))
    for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]: print(r['Name'][:100].ljust(100), r['Calls'], round(float(r['AverageNs']) / 1e3, 2))
except Exception as e: print('no stats', e)
