import csv, sys, collections
pat = sys.argv[1]
for d in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(d + "/out_counter_collection.csv")):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])] += 1
    for k, v in acc.items():
        for c, x in v.items():
            print(k, c, "%.4g per launch" % (x / cnt[(k, c)]), "launches", cnt[(k, c)])
