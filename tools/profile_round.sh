#!/bin/bash
# One measured round on the GPU box: bench line, rocprofv3 kernel stats of the same command, K4 HBM traffic
# (PMC, separate passes).  Writes gpurun_out/<tag>/...; copy what should be judged into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'tools/profile_round.sh r01_c'
set -e
tag=$1
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json
python3 bench.py --side-figures --no-cpu-baseline 2>/dev/null | tail -1 > $out/bench_side_figures.json
python3 bench.py --batch 256 --steps 5 --warmup 2 2>/dev/null | tail -1 > $out/bench_batch256.json
python3 bench.py --image16k --steps 5 --warmup 2 2>/dev/null | tail -1 > $out/bench_image16k.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --no-cpu-baseline --no-photographs --steps 40 --warmup 3 > $out/stats.log 2>&1
f=$(ls $out/stats/*kernel_stats.csv $out/stats/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $out/kernel_stats.csv
rm -rf $out/stats
head -12 $out/kernel_stats.csv
# HBM traffic of K4 on the SAME command the bench line comes from (the whole decode, not K4 alone): separate passes per counter
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-photographs > $out/pmc_$c.log 2>&1 || true
done
python3 - "$out" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + '/pmc_*/*/*counter_collection.csv') + glob.glob(out + '/pmc_*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'k_idct_colour_fast' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in agg.items()}
if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
    fetch = m['FETCH_SIZE'] * 1024 * 2   # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md)
    write = m['WRITE_SIZE'] * 1024
    json.dump({"kernel": "k_idct_colour_fast", "workload": "7680x4320 q75, inside the whole decode (bench.py --steps 5 --warmup 2 --no-cpu-baseline)", "FETCH_SIZE_KB": m['FETCH_SIZE'],
               "WRITE_SIZE_KB": m['WRITE_SIZE'], "fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write,
               "launches_averaged": len(agg.get('FETCH_SIZE', [])),
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile_round.sh); FETCH_SIZE doubled: on gfx950 it "
                       "reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section). Algorithmic bytes 298.6 MB + 6.2 MB of error bounds."},
              open(out + '/k4_traffic.json', 'w'), indent=1)
    print(open(out + '/k4_traffic.json').read())
PY
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
