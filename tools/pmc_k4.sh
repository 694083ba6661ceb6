#!/bin/bash
# PMC passes for K4 on the GPU box (counters only; no tracing domains besides --kernel-trace).
# usage: tools/pmc_k4.sh <outdir-under-gpurun_out> [bench args...]
set -e
out=gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --idct-only "${BARGS[@]}" > $out/$name.log 2>&1 || true
}
BARGS=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq3 SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fh:
    for k, d in agg.items():
        if 'idct' not in k: continue
        fh.write(k + '\n')
        for c, v in sorted(d.items()):
            fh.write('  %-28s mean %.4g  (n=%d)\n' % (c, sum(v) / len(v), len(v)))
print(open(out + '/summary.txt').read())
PY
