cd $GRAFT_REPO_ROOT
python tools/perf/cast_census2.py 2>&1 | grep -v "Warn\|warn\|amdgpu" | tail -60
python tools/perf/fill_census.py 2>&1 | grep -v "Warn\|warn\|amdgpu" | tail -70
