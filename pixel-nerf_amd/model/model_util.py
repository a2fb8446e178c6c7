"""make_mlp / make_encoder: src/model/model_util.py:5-28."""
from .encoder import SpatialEncoder
from .resnetfc import ResnetFC


def make_mlp(conf, d_in, d_latent=0, allow_empty=False, **kwargs):
    mlp_type = conf.get_string("type", "mlp")
    if mlp_type == "resnet":
        return ResnetFC.from_conf(conf, d_in, d_latent=d_latent, **kwargs)
    if mlp_type == "empty" and allow_empty:
        return None
    # type = mlp (ImplicitNet) is unreachable in the reference too (model_util.py:8 NameError)
    raise NotImplementedError("Unsupported MLP type")


def make_encoder(conf, **kwargs):
    enc_type = conf.get_string("type", "spatial")
    if enc_type == "spatial":
        return SpatialEncoder.from_conf(conf, **kwargs)
    if enc_type == "global":
        from .encoder import ImageEncoder
        return ImageEncoder.from_conf(conf, **kwargs)
    raise NotImplementedError("Unsupported encoder type")
