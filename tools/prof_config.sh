# usage: tools/prof_config.sh <config> <tag> [steps]   -> gpurun_out/<tag>/<config>_kernel_stats_single_stream.md
cfg=$1; tag=$2; steps=${3:-20}
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMDI_GROUPS=1 CMDI_PIPELINES=0 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag/prof_$cfg -- python bench.py --config $cfg --steps $steps --warmup 3 --no-cpu --no-pmc --no-f32 --no-roofline --precision f16x3 > gpurun_out/$tag/prof_$cfg.log 2>&1
python tools/rocpd_summary.py "$(find gpurun_out/$tag/prof_$cfg -name "*.db" | head -1)" gpurun_out/$tag/${cfg}_kernel_stats_single_stream.md "${ROUND_LABEL:-round 6}, CMDI_GROUPS=1 CMDI_PIPELINES=0: bench.py --config $cfg --steps $steps --warmup 3 (single stream)" > /dev/null 2>&1
python tools/rocpd_summary.py --by-grid "$(find gpurun_out/$tag/prof_$cfg -name "*.db" | head -1)" gpurun_out/$tag/${cfg}_by_grid.md > /dev/null 2>&1
rm -rf gpurun_out/$tag/prof_$cfg
