"""Drop-in for the reference's `tinygemm_lib` package: re-exports any4_amd's functional API and utils."""
