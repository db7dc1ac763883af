import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"][:70]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r["Dispatch_Id"])
for k, v in agg.items():
    n = len(cnt[k])
    print(k, "dispatches", n, {a: round(b / n) for a, b in v.items()})
