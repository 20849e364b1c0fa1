#!/bin/bash
# Run ON THE GPU BOX: everything DESIGN.md quotes -- the -m gpu suite, the rocprofv3 collections of every
# configuration, the bench lines.  Afterwards, in the build container:
#   python profiles/summarize.py r02 slice_cnc fused_cnc_seq fused_cnc_2q slice_l1 fused_l1_seq fused512_cnc_seq fused512_cnc fused_f64_seq fused_f64_2q fused_f64_chunk fused_f64_default generic_cnc
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/final/pytest.log
tail -4 gpurun_out/final/pytest.log
bash profiles/collect.sh slice_cnc
PNP_SLICE=0 PNP_FUSED_STREAMS=1 bash profiles/collect.sh fused_cnc_seq
PNP_SLICE=0 bash profiles/collect.sh fused_cnc_2q
bash profiles/collect.sh slice_l1 --solver l1
PNP_SLICE=0 PNP_FUSED_STREAMS=1 bash profiles/collect.sh fused_l1_seq --solver l1
PNP_FUSED_STREAMS=1 bash profiles/collect.sh fused512_cnc_seq --size 512 --batch 256
bash profiles/collect.sh fused512_cnc --size 512 --batch 256
PNP_FUSED_STREAMS=1 PNP_FUSED_CHUNK=-1 bash profiles/collect.sh fused_f64_seq --precision f64
PNP_FUSED_CHUNK=-1 bash profiles/collect.sh fused_f64_2q --precision f64
PNP_FUSED_STREAMS=1 bash profiles/collect.sh fused_f64_chunk --precision f64
bash profiles/collect.sh fused_f64_default --precision f64
bash profiles/collect.sh generic_cnc --generic
bash profiles/run_bench_lines.sh
