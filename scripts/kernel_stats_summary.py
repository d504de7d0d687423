import csv, sys, collections
f = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    n = r.get("Name") or r.get("Kernel_Name")
    agg[n[:70]][0] += int(r["Calls"]); agg[n[:70]][1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{t/1e3/steps:10.1f} us/step  {c:7d} calls  {t/c/1e3:8.2f} us/call  {100*t/tot:5.1f}%  {n}")
