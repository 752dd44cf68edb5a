mkdir -p gpurun_out/r3g
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMDI_GROUPS=1 CMDI_PIPELINES=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r3g/prof -- python bench.py --config c2 --steps 20 --warmup 3 --no-cpu --no-pmc --no-f32 --no-roofline --precision bf16x6 > gpurun_out/r3g/prof.log 2>&1
python tools/rocpd_summary.py "$(find gpurun_out/r3g/prof -name "*.db" | head -1)" gpurun_out/r3g/c2_bf16x6_kernel_stats_single_stream.md "round 3, bf16x6, CMDI_GROUPS=1 CMDI_PIPELINES=0: bench.py --config c2 --steps 20 --warmup 3 --precision bf16x6" > /dev/null 2>&1
rm -rf gpurun_out/r3g/prof
cut -c1-150 gpurun_out/r3g/c2_bf16x6_kernel_stats_single_stream.md | head -20
