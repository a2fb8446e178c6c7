"""src/render/__init__.py: `from render import NeRFRenderer`."""
from .nerf import NeRFRenderer  # noqa: F401
