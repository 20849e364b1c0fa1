# A/B of conv kernel variants on one box: bash profiles/variants.sh build kernels_conv.hip <name> "<-D flags>" first
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  lib=$R/build/variants/lib_$v.so; [ $v = base ] && lib=$R/pnp_admm_cnc_mri_amd/libpnpmri.so
  echo "$v: $(PNP_MRI_LIB=$lib python3 $R/profiles/experiments/probe_conv.py 2>/dev/null | grep -E '13 layers|hip mfma' | tr '\n' ' ')"
done
