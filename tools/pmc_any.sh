#!/bin/bash
# PMC passes over the full bench; summary for kernels matching $2 (regex). usage: tools/pmc_any.sh <outdir> <regex> [env KPEG_HIP_LIB]
set -e
out=gpurun_out/$1; pat=$2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/$name.log 2>&1 || true; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys, collections, re
out, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(out + '/sq1/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'].split('(')[0]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, d in agg.items():
    if not re.search(pat, k): continue
    print(k, 'calls', len(dur[k]), 'dur_us max %.1f mean %.1f' % (max(dur[k]), sum(dur[k]) / len(dur[k])))
    for c, v in sorted(d.items()):
        print('  %-24s max %.4g mean %.4g' % (c, max(v), sum(v) / len(v)))
PY
