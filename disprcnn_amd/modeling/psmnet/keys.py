"""State-dict layout of PSMNet (514 entries) without touching the GPU."""


def psmnet_state_template():
    from .stackhourglass import PSMNet
    return PSMNet(48, -48).state_dict()
