#!/bin/bash
# per-kernel time of one tool run under rocprofv3 (kernel trace + stats):  bash tools/kernel_breakdown.sh <forwards> <script> [args...]
# prints the 24 heaviest kernels with their time per forward
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kb && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -o kb -- python3 $R/"$@" > /dev/null 2>&1
python3 - "$N" <<'PY'
import csv, glob, sys
n = float(sys.argv[1])
f = glob.glob("/tmp/kb/**/kb_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total %.3f ms per forward" % (tot / 1e6 / n))
for r in rows[:24]:
    print("%-72s %5s calls %8.3f ms/fwd %5.1f %%" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6 / n, 100 * float(r["TotalDurationNs"]) / tot))
PY
