import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 8
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total ms/step", tot / steps / 1e6)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f"{int(r['TotalDurationNs'])/steps/1e6:7.3f} ms  x{int(r['Calls'])/steps:6.1f}  {r['Name'][:100]}")
