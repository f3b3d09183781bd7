"""Import shim: the parts of the reference's `facelib` that exist on the HIP path (face parsing only)."""
