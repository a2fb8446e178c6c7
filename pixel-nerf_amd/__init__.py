"""
pixelnerf_amd: MI355X-native (gfx950, hand-written HIP) pixelNeRF volume-rendering hot path
behind the reference's NeRFRenderer / PixelNeRFNet Python API.

Sub-packages mirror the reference layout (src/render, src/model, src/util) so that
`from pixelnerf_amd.render import NeRFRenderer`, `from pixelnerf_amd.model import make_model`
replace `from render import NeRFRenderer`, `from model import make_model`.
"""
__version__ = "0.1.0"
