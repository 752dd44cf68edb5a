mkdir -p gpurun_out/r4s
for f in ${KSTEP_BINS:-tools/probes/kstep_*}; do [ -x "$f" ] && case "$f" in *.hip) ;; *) timeout 60 $f;; esac; done > gpurun_out/r4s/${KSTEP_OUT:-kstep.txt} 2>&1
cat gpurun_out/r4s/${KSTEP_OUT:-kstep.txt}
