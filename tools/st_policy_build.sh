# experiment: variant libraries with cache policies on output stores / operand requests (gemm_h3.hpp h3_store_*)
cd diffusion-motion-inbetweening_amd/csrc
CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc -I ."
build() {  # name, flags: rebuilds the three translation units that contain the stores
  for u in gemm_h3 gemm_h3p attention_h3 elementwise; do $CC $2 -c $u.hip -o build/${u}_$1.o || exit 1; done
  objs=$(ls build/*.o | grep -v "_p[0-9]*\.o" | grep -v "/gemm_h3.o" | grep -v "/gemm_h3p.o" | grep -v "/attention_h3.o" | grep -v "/elementwise.o")
  hipcc --offload-arch=gfx950 -shared -fPIC -o libcondmdi_hip_$1.so $objs build/gemm_h3_$1.o build/gemm_h3p_$1.o build/attention_h3_$1.o build/elementwise_$1.o || exit 1
}
build p1 "-DCMDI_BWD_SC1=1" &
wait
ls -la libcondmdi_hip_p*.so
