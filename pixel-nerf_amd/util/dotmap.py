"""Minimal DotMap: what src/render/nerf.py uses from the `dotmap` package (attribute access,
auto-created empty children -- callers test `len(render_dict.fine) == 0`, train/train.py:201 --
and recursive toDict())."""


class DotMap(dict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = DotMap(v) if isinstance(v, dict) and not isinstance(v, DotMap) else v

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        if k not in self:
            self[k] = DotMap()
        return self[k]

    def __setattr__(self, k, v):
        self[k] = v

    def toDict(self):
        return {k: (v.toDict() if isinstance(v, DotMap) else v) for k, v in self.items()}
