"""Helper of test_model_gpu.py: losses and summed gradient of a launch of G mini-batches of the model_cfg2_multi.npz fixture in a fresh
process (the engine's switches are read once per process).  usage: python _multi_helper.py G out.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from _pkg import load_pkg  # noqa: E402
import test_model_gpu as T  # noqa: E402

G = int(sys.argv[1])
pkg = load_pkg()
gm, hp, cdl_o = T.multi_golden_state()
ctx = pkg._lib.Context(0)
cdl = T.to_model(pkg, ctx, hp, 200, cdl_o, arena=int((0.3 * G + 2) * (1 << 30)))
loss, flat = T.gpu_loss_grad(pkg, ctx, cdl, gm["codes"][: G * hp.batch_size], G)
np.savez(sys.argv[2], loss=loss, flat=flat)
