"""Per-layer comparison of in-situ runs with forced tile configurations (tools/perf/cfg_sweep.sh):
python tools/perf/cfg_sweep_join.py <dir> heur 14 15 ...   reads <dir>/shapes_heur.txt, <dir>/shapes_cfg14.txt, ..."""
import ast
import re
import sys

d, tags = sys.argv[1], sys.argv[2:]
runs = {}
for t in tags:
    name = "heur" if t == "heur" else f"cfg{t}"
    r = {}
    for line in open(f"{d}/shapes_{name}.txt"):
        m = re.match(r"\s+([\d.]+)\s+x\s*(\d+)\s+([\d.]+)\s+(?:[\d.]+\s+)?(\(.*\))\s+(\S.*)", line)   # (optional 4th column: us at 70 % of the peak)
        if m:
            key = ast.literal_eval(m.group(4))
            if key[0] == "fwd" and key[8] == 3:
                r[key] = (float(m.group(1)), m.group(5).strip())
    runs[t] = r
keys = sorted(runs["heur"], key=lambda k: -runs["heur"][k][0])
tot_h = sum(runs["heur"][k][0] for k in keys)
tot_b = sum(min(runs[t].get(k, (1e9,))[0] for t in tags) for k in keys)
print(f"3x3 fwd / data-gradient layers: heuristic {tot_h:.0f} us per iteration, best forced configuration per layer {tot_b:.0f} us")
for k in keys:
    h = runs["heur"][k][0]
    best = min(tags, key=lambda t: runs[t].get(k, (1e9,))[0])
    cells = "  ".join(f"{t}:{runs[t].get(k, (float('nan'),))[0]:6.0f}" for t in tags)
    print(f"{h - runs[best][k][0]:7.1f}  {str(k[1:]):58s} {cells}  best {best:5s} [{runs['heur'][k][1][:34]}]")
