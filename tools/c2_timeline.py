import csv, glob, sys
rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "m2v::" in r["Kernel_Name"]:
            n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("m2v::", "")[:14]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()
# take a window in the middle of the run where two queues alternate
mid = len(rows) // 2
t0 = rows[mid][0]
for s, e, n, q in rows[mid:mid + 40]:
    print("%9.1f us  +%7.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
