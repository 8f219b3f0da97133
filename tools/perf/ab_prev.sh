#!/bin/bash
# Same-box A/B of the working tree against a committed revision (box-to-box variation is +-5..10 %, same-box repeats
# agree to < 1 %): exports REV (default HEAD) into scratch/ab_prev, builds its library there (here, before gpurun ships
# the tree), and prints the command to run on the GPU box.
#   tools/perf/ab_prev.sh [REV]        then:  gpurun -- 'bash tools/perf/ab_run.sh'
set -e
REV=${1:-HEAD}
cd "$(dirname "$0")/../.."
rm -rf scratch/ab_prev && mkdir -p scratch/ab_prev
git archive "$REV" | tar -x -C scratch/ab_prev
(cd scratch/ab_prev && python -m layout2img_amd.build --force | tail -1)
