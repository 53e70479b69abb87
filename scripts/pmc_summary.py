"""summarise rocprofv3 --pmc CSVs: per kernel name, dispatch count and mean counter value."""
import csv, glob, os, sys, collections
root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for d in sorted(glob.glob(os.path.join(root, "*"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        print(d, "no counter csv:", [os.path.basename(x) for x in glob.glob(os.path.join(d, "**", "*"), recursive=True)][:6])
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    with open(files[0]) as f:
        rd = csv.DictReader(f)
        for row in rd:
            key = (row.get("Kernel_Name", "?")[:70], row.get("Counter_Name", "?"))
            agg[key][0] += 1
            agg[key][1] += float(row.get("Counter_Value", 0))
    print("==", os.path.basename(d))
    for (k, c), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print("  %-60s %-11s calls=%5d mean=%.6g total=%.6g" % (k, c, n, tot / n, tot))
