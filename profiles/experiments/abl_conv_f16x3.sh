# timing ablations of the f16x3 conv kernel (results wrong by design): which phase costs what on the full chip.
# build here:  bash profiles/experiments/abl_conv_f16x3.sh build      run on the GPU box:  bash profiles/experiments/abl_conv_f16x3.sh run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
V="WSAME NOXPRE NOWWRITE NOBAR NOEPI NOPUT"
if [ "$1" = build ]; then
  for v in $V; do bash $R/profiles/variants.sh build kernels_conv_f16x3.hip abl_$v "-DH3_ABL_$v" | tail -1; done
else
  cd $R
  for rep in 1 2; do
    for v in base $V; do
      if [ $v = base ]; then L=$R/pnp_admm_cnc_mri_amd/libpnpmri.so; else L=$R/build/variants/lib_abl_$v.so; fi
      echo "rep $rep $v: $(PNP_MRI_LIB=$L python3 profiles/experiments/probe_conv.py 64 128 128 1 f16x3 2>/dev/null | grep 'hip mfma')"
    done
  done
fi
