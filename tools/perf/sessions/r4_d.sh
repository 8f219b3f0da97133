cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dual.py -x -q -k forward_dual 2>&1 | tail -8
python -m pytest "tests/test_gpu_models.py::test_graph_replay_gradients_match_eager_at_full_size" -x -q 2>&1 | grep -v "^  \|Warn" | tail -30
echo "== dual"; bash tools/perf/timeline.sh 2>&1 | tail -16
echo "== single"; L2I_DUAL_D=0 bash tools/perf/timeline.sh 2>&1 | tail -16
