"""Sampler layer (mirror of the reference's ``diffusion`` package, sampling half only)."""
