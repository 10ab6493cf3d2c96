import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    print(r["Name"][:84].ljust(84), r["Calls"].rjust(5), f'{float(r["AverageNs"]) / 1e6:9.3f} ms', r["Percentage"].rjust(7))
