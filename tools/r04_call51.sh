set -e
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_more_gpu.py -x -q -m gpu -k "parity or fast_bound" 2>&1 | tail -3
python bench.py --steps 4 --warmup 1 2>/dev/null | cut -c1-260
