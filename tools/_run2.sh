cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
