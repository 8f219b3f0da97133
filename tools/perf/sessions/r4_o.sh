cd $GRAFT_REPO_ROOT
A="--no-cpu-baseline --no-g-forward --no-f32-mode --no-kernel-timer --steps 40"
run() { python bench.py $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'])"; }
run base
L2I_WGRAD_CAP=384 run wcap384; L2I_WGRAD_CAP=768 run wcap768; L2I_WGRAD_CAP=1024 run wcap1024
L2I_CONV_CFG=1384 run split384; L2I_CONV_CFG=1768 run split768; L2I_CONV_CFG=1256 run split256
run base
L2I_ROI_LIVE=60 run roilive60; L2I_NO_SMALL_HALO=0 run smallhalo0; L2I_NO_SMALL_HALO=3 run smallhalo3
