import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_row_receive" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("   %8.1f %8.1f %7.1f  q%s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Queue_Id"]))
print("   span %.1f us for %d kernels" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e3, len(rows)))
