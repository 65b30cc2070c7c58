"""libjxl_amd -- MI355X (gfx950) VarDCT decode back-end for libjxl.

Product code.  The HIP extension (csrc/ -> libjxl_hip.so) is mandatory: there
is no CPU fallback and nothing here imports oracle/.
"""
from .abi import (FrameParams, FrameInputs, LoopFilter, load_library,  # noqa: F401
                  library_path, make_params, JxlHipError)
from .decoder import VarDctDecoder  # noqa: F401
