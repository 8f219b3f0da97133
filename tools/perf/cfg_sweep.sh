#!/bin/bash
# In-situ sweep of forced tile configurations (on the GPU box): one shape_profile run per configuration, then the per-layer
# comparison (tools/perf/cfg_sweep_join.py). usage: tools/perf/cfg_sweep.sh "14 15 19 29" -> gpurun_out/cfg_sweep.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/perf/shape_profile.sh heur
for c in $1; do L2I_CONV_CFG=$c bash $R/tools/perf/shape_profile.sh cfg$c; done
python $R/tools/perf/cfg_sweep_join.py $R/gpurun_out heur $1 > $R/gpurun_out/cfg_sweep.txt
head -60 $R/gpurun_out/cfg_sweep.txt
