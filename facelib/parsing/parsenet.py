from codeformer_amd.facelib.parsing.parsenet import ConvLayer, NormLayer, ParseNet, ReluLayer, ResidualBlock  # noqa: F401
