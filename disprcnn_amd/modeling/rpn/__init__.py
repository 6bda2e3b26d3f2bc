from .stereo_rpn import StereoRPN, build_stereorpn  # noqa: F401
