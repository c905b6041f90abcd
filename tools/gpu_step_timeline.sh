cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/prof_tw; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_tw -o tw -- python $GRAFT_REPO_ROOT/tools/bench_mat_twist.py > /dev/null 2>&1 )
db=$(find /tmp/prof_tw -name "*.db" | head -1)
python tools/step_timeline.py $db 3 12
