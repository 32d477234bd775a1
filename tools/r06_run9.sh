cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 600 python tools/step_ab.py 2>&1 | tail -30
