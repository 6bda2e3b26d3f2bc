"""Development tool: stride-1 3x3x3 layers at small unit counts (Config B's quarter-resolution hourglass layers): kernel choices."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from disprcnn_amd import engine as E
import tools.exp_conv as X
for N in (4, 8, 16):
    for small in (True, False):
        E.WINO["small"] = small
        X.run(N, 64, 64, (6, 14, 14))
        X.run(N, 64, 64, (12, 28, 28))
        X.run(N, 32, 32, (24, 56, 56))
