from codeformer_amd.facelib.parsing import ParseNet, init_parsing_model  # noqa: F401
