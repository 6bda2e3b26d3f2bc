"""Case parameters of structures_golden.npz (shared by make_golden_structures.py and tests/test_structures.py)."""
# (name, source [H, W], destination (width, height))
RESIZE = [("up", (37, 53), (96, 80)), ("down", (120, 200), (64, 48)), ("same", (40, 40), (40, 40)), ("thin", (1, 17), (9, 5)),
          ("kitti_roi", (93, 141), (112, 112)), ("one", (8, 8), (1, 1))]
CROPS = [("inside", (50, 80), (10, 5, 60, 40)), ("past_edge", (50, 80), (60, 30, 100, 70)), ("frac", (50, 80), (3.4, 2.6, 20.5, 30.49)),
         ("empty", (50, 80), (10, 10, 10, 25))]
BOXES = [[10.2, 5.1, 90.7, 60.3], [60.0, 20.0, 200.0, 90.0], [150.5, 0.0, 319.2, 95.5], [0.0, 0.0, 0.0, 0.0], [300.0, 90.0, 319.0, 95.0]]
SIZE = (320, 96)
