db=$(find /tmp/prof_bench -name "*.db" | head -1)
for w in 12 20 31 47 60; do echo "== iteration $w"; python tools/rocprof_timeline.py $db $w | awk '{split($3,a,"="); if (a[2]+0 > 15) print}'; done
