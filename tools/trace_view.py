"""development aid: print a window of a rocprofv3 kernel trace (csv) as a per-queue timeline"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 60
base = int(rows[lo]["Start_Timestamp"])
last_end = {}
for r in rows[lo:lo + cnt]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name).split("(")[0][:34]
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    print("%9.2f us  q%-3s dur %7.2f  gap_same_q %7.2f  %s" % ((s - base) / 1e3, q, (e - s) / 1e3, gap, name))
    last_end[q] = e
