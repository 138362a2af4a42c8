"""Average duration of the single-step / multi-step k_chain launches in a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ch = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6 for r in rows if "k_chain<8, false" in r["Kernel_Name"]]
small = [d for d in ch if d < 5]
big = [d for d in ch if d > 5]
print("%s: single-step avg %.4f ms (n=%d), multi-step %.3f ms" % (sys.argv[1].split("/")[-1], sum(small) / max(len(small), 1), len(small), sum(big) / max(len(big), 1)))
