set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_zz_no_alloc.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -25
