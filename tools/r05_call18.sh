tag=r5p; mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for b in 2 10; do
CMDI_GROUPS=1 CMDI_PIPELINES=0 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/prof_b$b -- python bench.py --config c2 --batch $b --steps 50 --warmup 5 --no-cpu --no-pmc --no-f32 --no-roofline --no-graph-leg --precision f16x3 > gpurun_out/$tag/prof_b$b.log 2>&1
python tools/rocpd_summary.py "$(find gpurun_out/$tag/prof_b$b -name "*.db" | head -1)" gpurun_out/$tag/b${b}_kernel_stats_single_stream.md "round 5, CMDI_GROUPS=1 CMDI_PIPELINES=0: bench.py --config c2 --batch $b --steps 50 --warmup 5 (single stream)" > /dev/null 2>&1
rm -rf gpurun_out/$tag/prof_b$b
head -n 22 gpurun_out/$tag/b${b}_kernel_stats_single_stream.md | cut -c1-140
done
