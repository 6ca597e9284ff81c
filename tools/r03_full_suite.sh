R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03full
mkdir -p $O
cd $R
( timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu ) > $O/pytest_full.txt 2>&1
grep "passed\|failed\|error" $O/pytest_full.txt | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
