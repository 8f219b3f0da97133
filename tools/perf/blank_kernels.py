"""Context of the kernels rocprofv3 reports without a name, and of every idle gap > 50 us, in the last replayed iterations."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel")]
ends = adam[1::2]
lo, hi = ends[-3] + 1, ends[-1] + 1
print("columns:", list(rows[0].keys()))
maxend = int(rows[lo - 1]["End_Timestamp"])
for i in range(lo, hi):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - maxend) / 1e3
    if r["Kernel_Name"].strip() == "" or gap > 50:
        for j in (i - 2, i - 1, i, i + 1):
            q = rows[j]
            print(("  >> " if j == i else "     ") + f"t={(int(q['Start_Timestamp']) - int(rows[lo]['Start_Timestamp'])) / 1e3:10.1f} us dur={(int(q['End_Timestamp']) - int(q['Start_Timestamp'])) / 1e3:7.1f} "
                  f"queue={q.get('Queue_Id')} stream={q.get('Stream_Id', '')} grid={q.get('Grid_Size_X', q.get('Grid_Size'))} wg={q.get('Workgroup_Size_X', q.get('Workgroup_Size'))} lds={q.get('LDS_Block_Size')} name='{q['Kernel_Name'][:60]}'")
        print(f"     gap before >>: {gap:.1f} us")
    maxend = max(maxend, e)
