"""Factories the model conf selects its parts through (the reference keeps them in src/model/model_util.py:5-28): the `type` key of
an `mlp_*` / `encoder` conf subtree names the class, everything else in the subtree goes to that class's `from_conf`."""
from .encoder import ImageEncoder, SpatialEncoder
from .resnetfc import ResnetFC

_ENCODERS = {"spatial": SpatialEncoder, "global": ImageEncoder}


def make_mlp(conf, d_in, d_latent=0, allow_empty=False, **kwargs):
    """-> ResnetFC | None.  `type = resnet` is what every shipped conf uses; `empty` (mlp_fine only) means "no fine network";
    the reference's default `mlp` names a class (ImplicitNet) its module never imports, so it cannot be reached there either."""
    kind = conf.get_string("type", "mlp")
    if kind == "empty" and allow_empty:
        return None
    if kind != "resnet":
        raise NotImplementedError("Unsupported MLP type")
    return ResnetFC.from_conf(conf, d_in, d_latent=d_latent, **kwargs)


def make_encoder(conf, **kwargs):
    """-> SpatialEncoder (pixel-aligned feature grid, the default) | ImageEncoder (one global latent per image)."""
    cls = _ENCODERS.get(conf.get_string("type", "spatial"))
    if cls is None:
        raise NotImplementedError("Unsupported encoder type")
    return cls.from_conf(conf, **kwargs)
