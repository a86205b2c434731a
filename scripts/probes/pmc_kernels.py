"""Per-kernel averages of a rocprofv3 --pmc counter_collection csv: python pmc_kernels.py file.csv [name-filter]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in rows:
    k = re.sub(r"\(anonymous namespace\)::|void |\(gw\w*::\w+\)", "", r["Kernel_Name"])[:80]
    if flt and flt not in k:
        continue
    a = acc[k][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"]); a[1] += 1
for k, d in acc.items():
    print(k)
    for c, (s, n) in sorted(d.items()):
        print(f"   {c:28s} {s / n:16.1f}   (x{n})")
