from .conv import Conv1d, Conv2d, Conv3d, Deconv2d, Deconv3d  # noqa: F401
from .mlp import SharedMLP  # noqa: F401
