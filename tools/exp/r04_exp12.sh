cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e12; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu -s > $O/tests.log 2>&1; echo "rc $?" >> $O/tests.log
grep -v "^$" $O/tests.log | grep "full depth\|c4 at\|passed\|failed\|rc \|Error\|error" | tail -20
