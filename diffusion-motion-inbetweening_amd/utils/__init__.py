"""Host-side helpers mirroring the reference's ``utils`` package (hot-path subset)."""
