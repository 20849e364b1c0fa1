#!/bin/bash
# Run ON THE GPU BOX: SQ counters of the slice-resident kernel, one rocprofv3 pass per counter group (counters only).
#   bash profiles/pmc_sq.sh <name> [bench.py args...]   ->  gpurun_out/pmc_<name>/<group>/...counter_collection.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; shift
D=$R/gpurun_out/pmc_$NAME
rm -rf $D; mkdir -p $D
export PNP_BENCH_CACHE=/tmp/pnp_bench_inputs
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $D/counters.txt 2>&1 || true
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  echo "pass $i: $grp"
  rocprofv3 --pmc $grp --output-format csv -d $D/g$i -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $D/g$i.log 2>&1 || echo "  pass $i failed (see g$i.log)"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('$D/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_slice' in r.get('Kernel_Name', ''):
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, (v, n) in sorted(acc.items()):
    print('%-28s per launch %.4g  (%d records)' % (k, v / max(n, 1), n))
PY
