set -e
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
