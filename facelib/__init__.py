"""Import shim: the parts of the reference's `facelib` that exist here (face parsing, RetinaFace detection, the restoration helper)."""
