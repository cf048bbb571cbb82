#!/bin/bash
# on the GPU box: rocprofv3 kernel stats of one bench.py run, one line per kernel (name, calls, average us)
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/kst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kst -o b -- python $R/bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-speed-mode "$@" > /tmp/kst.log 2>&1
python - <<'PY'
import csv
for r in csv.DictReader(open('/tmp/kst/b_kernel_stats.csv')):
    n = r['Name'].split('(')[0].replace('void ', '').replace('adanerf::', '')
    if float(r['Percentage']) > 0.05: print('%-48s calls %4s avg %9.1f us  %5.2f%%' % (n[:48], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
grep '^{' /tmp/kst.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FPS %.1f' % d['value'], d['stage_ms_per_frame'])"
