"""Mirror of the reference package of the same name: thin callers of the TAV path kernels."""
