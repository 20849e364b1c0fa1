# round 6: HBM-side traffic of the 64 -> 64 conv3x3 layer at [640, 64, 128, 128] split -> split, narrow (PNP_CONV_WIDE=0) and wide (1) kernel:
# FETCH_SIZE and WRITE_SIZE in separate counters-only passes (MI355X_MICROARCH.md: FETCH_SIZE counts half of a wide coalesced read: x 2; units of 32 B
# are converted by rocprofv3's derived counter?  -- the script prints raw values and the guide's correction)
set -e
R=$GRAFT_REPO_ROOT
D=$R/gpurun_out/pmc_traffic_r06
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  export PNP_CONV_WIDE=$V
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $D/v$V/$C -- python3 $R/profiles/experiments/probe_conv_fmt.py 640 5 > $D/v$V.$C.log 2>&1 || echo "variant $V $C failed"
  done
done
python3 - <<PY
import csv, glob, collections
for V, kn in ((0, 'k_conv3x3_c64_h3'), (1, 'k_conv3x3_h3w')):
    for C in ('FETCH_SIZE', 'WRITE_SIZE'):
        vals = []
        for f in glob.glob('$D/v%d/%s/**/*counter_collection.csv' % (V, C), recursive=True):
            vals += [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if kn in r.get('Kernel_Name', '') and r['Counter_Name'] == C]
        if vals:
            v = sum(vals) / len(vals)
            print('%-18s %-11s per launch (avg of %d): raw %.4g  -> x 1024 B = %.1f MB%s' % (kn, C, len(vals), v, v * 1024 / 1e6, '  (x 2 per the guide: %.1f MB)' % (2 * v * 1024 / 1e6) if C == 'FETCH_SIZE' else ''))
print('algorithmic: 640 x 128 x 128 x 256 B = 2684.4 MB read (+ halo: x 1.41 for 8 x 16 tiles, x 1.27 for 16 x 16), 2684.4 MB written, 147 KB of weights')
PY
