"""Denoiser layer (mirror of the reference's ``model`` package: MDM trans_enc + CFG wrapper)."""
