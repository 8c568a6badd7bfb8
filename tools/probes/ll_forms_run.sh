cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_deep_ep_gpu.py -x -q -k "launch_forms" 2>&1 | tail -4
bash tools/probes/ll_forms_ab.sh
