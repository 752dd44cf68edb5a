# usage: bash tools/rebuild.sh [--probes-too]  — rebuild libcondmdi_hip.so (and the instrumented probes library) from anywhere
cd "$(dirname "$0")/.." || exit 1
python diffusion-motion-inbetweening_amd/build.py 2>&1 | tail -3
[ "$1" = "--probes-too" ] && python diffusion-motion-inbetweening_amd/build.py --probes 2>&1 | tail -3
ls -la --time-style=+%H:%M:%S diffusion-motion-inbetweening_amd/csrc/libcondmdi_hip.so diffusion-motion-inbetweening_amd/csrc/libcondmdi_hip_probes.so | awk '{print $6, $7}'
date +%H:%M:%S
