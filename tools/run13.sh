cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/test_gpu_robustness.py -q -x 2>&1 | tail -30 ) 2>&1 | tail -34
