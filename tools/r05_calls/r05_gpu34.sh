cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
PST_JIT_DEBUG=1 timeout 1300 python -m pytest tests -m gpu -q -s --tb=line > /tmp/suite.log 2>&1
grep -a -c "compilation failed" /tmp/suite.log
grep -a -n -A25 "pst jit\] compilation failed" /tmp/suite.log | cut -c1-300 | head -150 > gpurun_out/r05/jit_failures.txt
tail -5 /tmp/suite.log | cut -c1-300
grep -a -n "loading the compiled plan failed" /tmp/suite.log | head -5
