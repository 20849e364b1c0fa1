mkdir -p gpurun_out/r06; export PNP_CONV_WIDE=1
for lib in h3wprof p_oob p_nost; do for fmt in 5 0; do echo "== $lib fmt $fmt"; PNP_MRI_LIB=build/variants/lib_$lib.so timeout -k 10 120 python3 profiles/experiments/prof_conv_f16x3_wide_phases.py 640 64 128 128 $fmt 2>&1 | grep -E "ms \(instr|taps|epilogue|B_1|B_0"; done; done
