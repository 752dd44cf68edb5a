mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMDI_GROUPS=1 CMDI_PIPELINES=0 rocprofv3 --kernel-trace --stats -d gpurun_out/r2/prof_c2 -- python bench.py --steps 40 --warmup 5 --no-cpu --no-pmc --no-f32 --no-roofline > gpurun_out/r2/prof_c2.log 2>&1
python tools/rocpd_summary.py "$(find gpurun_out/r2/prof_c2 -name "*.db" | head -1)" gpurun_out/r2/c2_kernel_stats_single_stream.md "round 2 (final), LN folded, CMDI_GROUPS=1 CMDI_PIPELINES=0: bench.py --steps 40 --warmup 5 (c2)" > /dev/null 2>&1
rm -rf gpurun_out/r2/prof_c2
cat gpurun_out/r2/c2_kernel_stats_single_stream.md | cut -c1-200
python bench.py --config c4 --no-cpu > gpurun_out/r2/bench_c4.json 2> gpurun_out/r2/bench_c4.err; tail -c 300 gpurun_out/r2/bench_c4.json
