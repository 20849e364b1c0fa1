export PNP_BENCH_CACHE=/tmp/pb
run() { env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-f64-record $ARGS 2>/dev/null | grep -o '"value": [0-9.]*' | cut -d' ' -f2; }
for ARGS in "--steps 20 --warmup 5" "--steps 100"; do
 for rep in 1 2 3; do
  echo "[$ARGS] rep $rep base $(run PNP_X=0)"
  for k in 1 2 3 4; do
    K1=$k; if [ "$ARGS" = "--steps 100" ]; then K1=$((k*4)); fi
    echo "[$ARGS] rep $rep split K1=$K1 $(run PNP_MRI_LIB=build/variants/lib_split.so PNP_SLICE_SPLIT_K1=$K1)"
  done
 done
done
