"""disprcnn_amd -- MI355X-native instance-disparity hot path of Disp R-CNN.

Import paths mirror the reference package (``disprcnn.modeling.psmnet.stackhourglass.PSMNet``,
``disprcnn.layers.ROIAlign``, ``disprcnn.utils.loss_utils.PSMLoss``) under ``disprcnn_amd``.
"""
__version__ = "0.1.0"
