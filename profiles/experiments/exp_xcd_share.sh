# Round 4: is the odd workgroups' slowness a smaller SHARE of a saturated memory system, or a limit of their own?
cd $GRAFT_REPO_ROOT/profiles/micro
for args in "4 256 50 0 0 0" "4 256 50 0 1 0" "4 256 50 0 2 0" "4 256 50 25 0 0" "4 256 50 25 1 0" "4 256 50 25 2 0" "4 256 50 25 0 2" "4 256 50 25 0 4" "4 256 50 25 0 8" "4 256 50 0 0 4" "4 256 50 0 0 8"; do
  ./slice_stride $args | tail -1
done
