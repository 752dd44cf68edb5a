# usage: bash tools/variant_build.sh <name> "<-D flags>" [units...]  -> csrc/libcondmdi_hip_<name>.so (CMDI_LIB_VARIANT=<name>)
# Rebuilds the listed translation units (default: the GEMM family + unet) with extra -D flags and links them with the
# product build's other objects (VARIANT_BASE=build_probes: with the instrumented build's — cycle stamps, CMDI_PROBES_LIB tools; add
# -DCMDI_PROBES to the flags).  Same-box A/B of library variants: tools/ab_variants.sh.
base=${VARIANT_BASE:-build}
name=$1; flags=$2; shift 2
units=${@:-gemm_h3 gemm_h3p unet}
cd "$(dirname "$0")/../diffusion-motion-inbetweening_amd/csrc" || exit 1
CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc -I ."
objs=""
for o in $base/*.o; do
  b=$(basename $o .o); skip=0
  case "$b" in *_v_*) skip=1;; esac
  for u in $units; do [ "$b" = "$u" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for u in $units; do
  extra=""; case "$u" in sampler|postprocess) extra="-ffp-contract=off";; gemm_h3w) extra="-fno-slp-vectorize";; esac
  $CC $flags $extra -c $u.hip -o $base/${u}_v_$name.o || exit 1
  objs="$objs $base/${u}_v_$name.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-rdc -o libcondmdi_hip_$name.so $objs || exit 1
ls -la libcondmdi_hip_$name.so
