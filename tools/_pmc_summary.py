"""Per-kernel sums of the rocprofv3 --pmc counters in a counter_collection.csv (one row per dispatch and counter)."""
import csv, glob, sys, collections, json
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:70]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'], k)
    if key not in seen: seen.add(key); calls[k] += 1
out = {k: dict(calls=calls[k], **{c: v for c, v in acc[k].items()}) for k in acc}
top = sorted(out.items(), key=lambda kv: -(kv[1].get('GRBM_GUI_ACTIVE', 0) + kv[1].get('SQ_WAVE_CYCLES', 0)))[:14]
print(json.dumps(dict(top), indent=1))
