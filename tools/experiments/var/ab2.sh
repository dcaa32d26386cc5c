cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_split_accuracy.py -q 2>&1 | tail -12
