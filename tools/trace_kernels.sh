#!/bin/bash
# per-launch kernel durations of one bench run (rocprofv3 kernel trace) -> gpurun_out/trace/summary.txt
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf gpurun_out/trace; mkdir -p gpurun_out/trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/trace/bench.log 2>&1
f=$(ls gpurun_out/trace/*kernel_trace.csv gpurun_out/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY' > gpurun_out/trace/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:60]))
PY
rm -f gpurun_out/trace/*.csv gpurun_out/trace/*/*.csv
cat gpurun_out/trace/summary.txt
