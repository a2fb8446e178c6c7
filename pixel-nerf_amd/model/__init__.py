"""src/model/__init__.py:1-11."""
from .models import PixelNeRFNet


def make_model(conf, *args, **kwargs):
    model_type = conf.get_string("type", "pixelnerf")
    if model_type == "pixelnerf":
        return PixelNeRFNet(conf, *args, **kwargs)
    raise NotImplementedError("Unsupported model type", model_type)
