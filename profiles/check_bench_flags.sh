#!/bin/bash
# Run ON THE GPU BOX: bench.py flag combinations with tiny step counts -- every one must print exactly one JSON line (rc 0).
cd $GRAFT_REPO_ROOT
for args in "--solver l1 --size 512 --batch 8 --steps 2 --warmup 1 --cpu-budget 0.3" "--generic --batch 16 --steps 2 --warmup 1 --cpu-budget 0.3" "--precision f64 --size 512 --batch 4 --steps 2 --warmup 1 --cpu-budget 0.3" "--precision f64 --generic --batch 8 --steps 2 --warmup 1 --cpu-budget 0.3" "--batch 700 --steps 3 --warmup 1 --cpu-budget 0.3" "--gpus 2 --rehearse-gloo --size 512 --batch 8 --steps 2 --warmup 1" "--solver l1 --batch 64 --steps 3 --warmup 0 --cpu-budget 0.3"; do
  out=$(timeout -k 10 200 python bench.py $args 2>/tmp/err.txt); rc=$?
  echo "[$args] rc=$rc lines=$(echo "$out" | grep -c '^{') $(echo "$out" | python3 -c "import sys,json
try:
    j=json.loads(sys.stdin.read()); print('value %.1f path %s dtype %s parity %s' % (j['value'], j['config']['path'], j['dtype'], (j.get('parity') or {}).get('rel_l2_vs_oracle')))
except Exception as e: print('PARSE FAIL', e)")"
  [ $rc -ne 0 ] && tail -3 /tmp/err.txt
done
exit 0
