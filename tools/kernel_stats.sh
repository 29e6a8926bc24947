#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command; the per-kernel summary lands in <outdir>/<tag>_kernel_stats.csv.
#   tools/kernel_stats.sh <outdir> <tag> <seconds> python3 script.py args...      (the program itself after the tag: no wrappers)
out=$1; tag=$2; limit=$3; shift 3
mkdir -p "$out"
repo=${GRAFT_REPO_ROOT:-$(pwd)}
case "$out" in /*) ;; *) out="$repo/$out";; esac
work=$(mktemp -d /tmp/kstats_XXXXXX)
( TMPDIR=/tmp timeout -k 10 "$limit" rocprofv3 --kernel-trace --stats --output-format csv -d "$work" -- "$@" > "$out/$tag.stdout" 2> "$out/$tag.stderr" )
rc=$?
echo "[$tag] rc=$rc"
f=$(find "$work" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then cp "$f" "$out/${tag}_kernel_stats.csv"; python3 - "$out/${tag}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls %6s  avg %9.2f us  total %6.2f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
else echo "[$tag] no kernel_stats.csv under $work"; fi
rm -rf "$work"
exit $rc
