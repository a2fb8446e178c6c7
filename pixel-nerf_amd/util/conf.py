"""dict-backed stand-in for a pyhocon ConfigTree: the renderer / model constructors only use
conf.get_{bool,int,float,string,list}(key, default) and conf["sub"] (src/render/nerf.py:340-352,
src/model/models.py:21-73, src/model/resnetfc.py:186-198).  A real pyhocon tree works as well."""


class Conf(dict):
    def _get(self, key, default=None):
        return self[key] if key in self else default

    get_bool = get_int = get_float = get_string = get_list = get = _get

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        return Conf(v) if isinstance(v, dict) and not isinstance(v, Conf) else v


def default_model_conf():
    """conf/default.conf:3-48 merged with conf/default_mv.conf:3-22: the one model shape every
    shipped experiment resolves to."""
    mlp = dict(type="resnet", n_blocks=5, d_hidden=512, combine_layer=3, combine_type="average")
    return Conf(
        use_encoder=True, use_global_encoder=False, use_xyz=True, canon_xyz=False,
        use_code=True, code=dict(num_freqs=6, freq_factor=1.5, include_input=True),
        use_viewdirs=True, use_code_viewdirs=False,
        mlp_coarse=dict(mlp), mlp_fine=dict(mlp),
        encoder=dict(backbone="resnet34", pretrained=False, num_layers=4),
    )


def default_renderer_conf():
    """conf/default.conf:49-60."""
    return Conf(n_coarse=64, n_fine=32, n_fine_depth=16, depth_std=0.01, sched=[], white_bkgd=True)
