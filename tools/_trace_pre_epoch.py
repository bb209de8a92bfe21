import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find the last vectorise2 launch with edits (the last timed step) and print the 40 kernels before and 6 after
idx = [i for i, r in enumerate(rows) if 'vectorise2_kernel' in r['Kernel_Name']]
i0 = idx[-2] if len(idx) > 1 else idx[-1]
t0 = int(rows[max(i0 - 30, 0)]['Start_Timestamp'])
for r in rows[max(i0 - 30, 0):i0 + 12]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f}  {r['Kernel_Name'][:90]}")
