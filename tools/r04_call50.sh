set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_more_gpu.py -x -q -m gpu -k "frame_ahead_steady" 2>&1 | tail -5
