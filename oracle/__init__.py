"""CPU oracle of the CondMDI sampling hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / reported baseline; the product package
(``diffusion-motion-inbetweening_amd``) never does.

Contents
    weights.py           deterministic (numpy PCG64) MDM state dicts shared by fixtures and tests
    mdm_oracle.py        numpy fp32 restatement of MDM trans_enc forward + CFG + its input-VJP
    diffusion_oracle.py  numpy restatement of schedules, respacing, p_sample / ddim_sample, the
                         imputation / reconstruction-guidance branches, Philox4x32-10 + Box-Muller
    ref_shims.py         imports the REAL reference from /root/reference (this container only) —
                         used by tests/golden/make_golden.py to pin the restatement

Parity status: the reference ships NO tests or golden vectors for this path.
The oracle is pinned instead against outputs of the reference itself, run here on CPU and committed
as tests/golden/*.npz by tests/golden/make_golden.py.
"""
