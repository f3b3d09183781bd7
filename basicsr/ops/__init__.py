"""Import shim: `basicsr.ops.*` of the reference resolves to codeformer_amd.bundled (HIP, inference only)."""
