"""HIP-backed pieces of the reference's `facelib` package (only what runs on the GPU: the face-parsing network)."""
