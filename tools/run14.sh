cd $GRAFT_REPO_ROOT
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -12 ) 2>&1 | tail -16
