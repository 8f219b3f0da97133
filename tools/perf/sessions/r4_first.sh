cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-g-forward --steps 40"
python bench.py $A > gpurun_out/r4a_b32.json 2>gpurun_out/r4a_b32.err
python bench.py $A --batch 64 > gpurun_out/r4a_b64.json 2>gpurun_out/r4a_b64.err
python bench.py $A --batch 16 > gpurun_out/r4a_b16.json 2>gpurun_out/r4a_b16.err
L2I_OVERLAP=0 python bench.py $A > gpurun_out/r4a_b32_noov.json 2>gpurun_out/r4a_b32_noov.err
for f in b32 b64 b16 b32_noov; do python - <<PY
import json
d=json.loads(open("gpurun_out/r4a_$f.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("$f", d["value"], d["ms_per_step"], r["frac"], r["wgrad_frac"], r["launches_per_step"], d.get("eager"))
PY
done
