#!/bin/bash
# Run ON THE GPU BOX: the data movement of a hypothetical 4-workgroup-per-slice 512x512 kernel (profiles/micro/cluster_exchange.hip,
# DESIGN.md 4.4): state phase only / exchanges only / both, with and without memory-free pauses, clusters on one XCD or across XCDs;
# then the fabric bytes of the exchange-only and the full pattern (separate rocprofv3 --pmc passes).
cd $GRAFT_REPO_ROOT/profiles/micro
O=$GRAFT_REPO_ROOT/gpurun_out/cluster_exchange; mkdir -p $O
for same in 1 0; do for pause in 0 15; do for mode in 0 1 2; do timeout -k 5 60 ./cluster_exchange $mode 50 $pause $same | tail -1; done; done; done > $O/times.jsonl 2>&1
cat $O/times.jsonl
cd /tmp && export TMPDIR=/tmp
for mode in 1 2; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 120 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${mode}_$ctr -- $GRAFT_REPO_ROOT/profiles/micro/cluster_exchange $mode 50 15 1 > $O/pmc_${mode}_$ctr.log 2>&1
  done
done
python3 - <<PY
import csv, glob
for mode in (1, 2):
    out = {}
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        vals = []
        for f in glob.glob('$O/pmc_%d_%s/**/*counter_collection.csv' % (mode, ctr), recursive=True):
            for r in csv.DictReader(open(f)):
                if 'k_cluster' in r['Kernel_Name'] and r['Counter_Name'] == ctr:
                    vals.append(float(r['Counter_Value']))
        out[ctr] = sum(vals) / max(len(vals), 1)
    rd, wr = 2 * out['FETCH_SIZE'] * 1024, out['WRITE_SIZE'] * 1024          # KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B
    print('mode %d: fabric bytes per workgroup-iteration: read %.3f MB  write %.3f MB (256 workgroups x 50 iterations per launch)' % (mode, rd / 256 / 50 / 1e6, wr / 256 / 50 / 1e6))
PY
